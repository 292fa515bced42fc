// tests/hip/div3_check.hip -- does div3() (wayverb_amd/csrc/device_common.hip.h) equal the hardware's
// IEEE division by 3, bit for bit?  Built and run by tests/test_gpu_div3.py.
//   float : every one of the 2^32 bit patterns
//   double: 2^34 patterns -- hashed significands under every exponent (subnormals, inf / nan included),
//           plus +-4096 ulps around every power of two
//           plus the LOW RANGE, where quotients are subnormal or in the lowest normal binade and the proof of
//           device_common.hip.h argues in units of 2^-1074 instead of relative errors: every pattern below 2^32 of
//           both signs, and under each of the exponent fields 0..3 2^30 hashed significands plus +-65536 patterns
//           around the significands 0, 2^51 (where x / 3 crosses a binade) and 2^52 - 1
// NaN results compare equal when both are NaN with the same payload class (quiet); everything else bitwise.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../wayverb_amd/csrc/device_common.hip.h"

__global__ void check_float(unsigned long long* bad, unsigned* example) {
    const uint64_t n = 1ull << 32;
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const unsigned a = __float_as_uint(x / 3.0f), b = __float_as_uint(wv::div3(x));
        if (a != b) {
            ++mine;
            *example = (unsigned)i;
        }
    }
    if (mine) atomicAdd(bad, mine);
}

__global__ void check_double(unsigned long long* bad, unsigned long long* example, uint64_t n) {
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t h = i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        uint64_t u;
        if ((i & 3) == 0) {
            u = h;                                                                            // anything
        } else if ((i & 3) == 1) {
            u = (h & 0x800FFFFFFFFFFFFFull) | ((uint64_t)((i >> 2) % 2048) << 52);            // every exponent
        } else {
            const uint64_t e = (i >> 2) % 2047, k = (i >> 13) % 8192;                         // around powers of two
            u = ((e << 52) + k - 4096) | (h & 0x8000000000000000ull);
        }
        const double x = __longlong_as_double((long long)u);
        const uint64_t a = (uint64_t)__double_as_longlong(x / 3.0), b = (uint64_t)__double_as_longlong(wv::div3(x));
        if (a != b) {
            ++mine;
            *example = u;
        }
    }
    if (mine) atomicAdd(bad, mine);
}

__global__ void check_double_low(unsigned long long* bad, unsigned long long* example) {
    const uint64_t n_small = 1ull << 33, n_hashed = 4ull << 30, n_window = 4ull * 3 * (1ull << 17);
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_small + n_hashed + n_window;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t u;
        if (i < n_small) {
            u = (i >> 1) | ((i & 1) << 63);
        } else if (i < n_small + n_hashed) {
            const uint64_t k = i - n_small;
            uint64_t h = k * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 32;
            u = (h & 0x800FFFFFFFFFFFFFull) | ((k & 3) << 52);
        } else {
            const uint64_t k = i - n_small - n_hashed;
            const uint64_t e = k & 3, which = (k >> 2) % 3, off = (k >> 2) / 3;  // off < 2^17
            const uint64_t centre = which == 0 ? 0 : (which == 1 ? (1ull << 51) : ((1ull << 52) - 1));
            u = (((e << 52) | centre) + off - 65536) & 0x7FFFFFFFFFFFFFFFull;
        }
        const double x = __longlong_as_double((long long)u);
        const uint64_t a = (uint64_t)__double_as_longlong(x / 3.0), b = (uint64_t)__double_as_longlong(wv::div3(x));
        if (a != b) {
            ++mine;
            *example = u;
        }
    }
    if (mine) atomicAdd(bad, mine);
}

int main() {
    unsigned long long *bad, h_bad[3] = {0, 0, 0}, *ex64, h_ex64[2] = {0, 0};
    unsigned *ex32, h_ex32 = 0;
    if (hipMalloc((void**)&bad, 24) != hipSuccess) {
        printf("no HIP device\n");
        return 2;
    }
    hipMalloc((void**)&ex64, 16);
    hipMalloc((void**)&ex32, 4);
    hipMemset(bad, 0, 24);
    hipLaunchKernelGGL(check_float, dim3(256 * 16), dim3(256), 0, 0, bad, ex32);
    hipLaunchKernelGGL(check_double, dim3(256 * 16), dim3(256), 0, 0, bad + 1, ex64, 1ull << 34);
    hipLaunchKernelGGL(check_double_low, dim3(256 * 16), dim3(256), 0, 0, bad + 2, ex64 + 1);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipMemcpy(h_bad, bad, 24, hipMemcpyDeviceToHost);
    hipMemcpy(&h_ex32, ex32, 4, hipMemcpyDeviceToHost);
    hipMemcpy(h_ex64, ex64, 16, hipMemcpyDeviceToHost);
    printf("float: %llu mismatches of 2^32 (e.g. 0x%08x)\ndouble: %llu mismatches of 2^34 (e.g. 0x%016llx)\n", h_bad[0], h_ex32,
           h_bad[1], h_ex64[0]);
    printf("double, low range: %llu mismatches of 2^33 + 2^32 + 3 * 2^19 (e.g. 0x%016llx)\n", h_bad[2], h_ex64[1]);
    return (h_bad[0] || h_bad[1] || h_bad[2]) ? 1 : 0;
}
