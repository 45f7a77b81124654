// Exhaustive check of bt_device.hpp's "same bits, fewer issue slots" helpers against the compiler's own IEEE expansions:
//   sqrt_bt(x) == sqrtf(x) and rsqrt_bt(x) == 1.0f / sqrtf(x) for ALL 2^32 bit patterns of x (NaN == NaN by class),
//   div_refined(p, q, refined_rcp(q)) == p / q for 2^32 pseudo-random (p, q) with both magnitudes in [2^-30, 2^30),
//   sincos_small_bt(x) == sincos_bt(x) for all 2^32 bit patterns.
// Build and run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Ibendy_tracer_amd/csrc -Iinclude \
//       tools/exact_math_check.hip -o gpurun_out/exact_math_check && gpurun_out/exact_math_check
#include <hip/hip_runtime.h>

#include <cstdio>

#include "bt_device.hpp"

namespace {
__device__ bool same(float a, float b) {
    const uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    return x == y || (a != a && b != b);
}
}  // namespace

__global__ void check(unsigned long long *bad, uint32_t *first_bad) {
    const uint64_t n = 1ull << 32;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)i;
        const float x = __uint_as_float(b);
        if (!same(sqrt_bt(x), sqrtf(x))) { atomicAdd(&bad[0], 1ull); atomicMin(&first_bad[0], b); }
        if (!same(rsqrt_bt(x), 1.0f / sqrtf(x))) { atomicAdd(&bad[1], 1ull); atomicMin(&first_bad[1], b); }
        // (p, q): two hashes of i, exponents folded into [-30, 30)
        uint32_t h = b * 0x9e3779b9u; h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13;
        uint32_t g = (b ^ 0x5bd1e995u) * 0xc2b2ae35u; g ^= g >> 16; g *= 0x27d4eb2fu; g ^= g >> 15;
        const float p = __uint_as_float((h & 0x807fffffu) | ((97u + (h >> 23) % 60u) << 23));
        const float q = __uint_as_float((g & 0x807fffffu) | ((97u + (g >> 23) % 60u) << 23));
        if (!same(div_refined(p, q, refined_rcp(q)), p / q)) { atomicAdd(&bad[2], 1ull); atomicMin(&first_bad[2], b); }
        // the widest window any caller gates on (bt_api.cpp: clip_max <= 2^60, |q| > 1e-5; the march's rel / size and the sphere
        // normals: |rel| in [2^-60, 2^60], size in [2^-20, 2^20]): |p| in [2^-60, 2^61), |q| in [2^-20, 2^20)
        const float pw = __uint_as_float((h & 0x807fffffu) | ((67u + (h >> 23) % 121u) << 23));
        const float qw = __uint_as_float((g & 0x807fffffu) | ((107u + (g >> 23) % 40u) << 23));
        if (!same(div_refined(pw, qw, refined_rcp(qw)), pw / qw)) { atomicAdd(&bad[4], 1ull); atomicMin(&first_bad[4], b); }
        float s0, c0, s1, c1;
        sincos_bt(x, s0, c0);
        sincos_small_bt(x, s1, c1);
        if (!same(s0, s1) || !same(c0, c1)) { atomicAdd(&bad[3], 1ull); atomicMin(&first_bad[3], b); }
    }
}

int main() {
    unsigned long long *bad;
    uint32_t *first;
    if (hipMalloc(&bad, 5 * sizeof(*bad)) != hipSuccess || hipMalloc(&first, 5 * sizeof(*first)) != hipSuccess) return 2;
    hipMemset(bad, 0, 5 * sizeof(*bad));
    hipMemset(first, 0xff, 5 * sizeof(*first));
    check<<<256 * 32, 256>>>(bad, first);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 2; }
    unsigned long long h[5];
    uint32_t f[5];
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(f, first, sizeof(f), hipMemcpyDeviceToHost);
    const char *name[5] = {"sqrt_bt vs sqrtf (all 2^32 inputs)", "rsqrt_bt vs 1/sqrtf (all 2^32 inputs)",
                           "div_refined vs p/q (2^32 pairs, |p|,|q| in [2^-30, 2^30))", "sincos_small_bt vs sincos_bt (all 2^32 inputs)",
                           "div_refined vs p/q (|p| in [2^-60, 2^61), |q| in [2^-20, 2^20))"};
    int rc = 0;
    for (int k = 0; k < 5; ++k) {
        std::printf("%-66s mismatches %llu", name[k], h[k]);
        if (h[k]) { std::printf("  first at bits 0x%08x", f[k]); rc = 1; }
        std::printf("\n");
    }
    return rc;
}
