// Microbenchmark: rate of scattered float atomic adds by memory scope and XCD placement.
// Question: do workgroup-scope atomics (sc0) execute in the XCD's L2 instead of memory-side,
// and how fast, when every address is only ever touched from ONE XCD?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: agent scope, any block -> any region     MODE 1: workgroup scope, same pattern (NOT coherent: timing only)
// MODE 2: workgroup scope, block adds only into the region owned by ITS XCD (coherent by construction)
// MODE 3: agent scope with the XCD-owned pattern (for comparison)
// PAIR: each pair of lanes targets the two floats of one 8-byte row (like C = 2)
template <int MODE>
__global__ __launch_bounds__(256) void k_atomics(float *table, uint32_t rows_per_region, uint32_t n_regions,
                                                 uint32_t iters, uint32_t *xcd_hist)
{
    const uint32_t xcd = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcd_hist[xcd], 1u);
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t region = (MODE >= 2) ? (xcd % n_regions) : (blockIdx.x % n_regions);
    float *base = table + (size_t)region * rows_per_region * 2;
    for (uint32_t i = 0; i < iters; i++) {
        const uint32_t row = hash32(gid * 977u + i * 0x9e3779b9u) % rows_per_region;
        float *p = base + (size_t)row * 2;
        if (MODE == 0 || MODE == 3) {
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

template <int MODE>
static void run(const char *name, float *table, uint32_t rows, uint32_t regions, uint32_t blocks, uint32_t iters,
                uint32_t *hist_d, bool check)
{
    CHECK(hipMemset(table, 0, (size_t)rows * regions * 8));
    CHECK(hipMemset(hist_d, 0, 64));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k_atomics<MODE><<<blocks, 256>>>(table, rows, regions, 1, hist_d);   // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(table, 0, (size_t)rows * regions * 8));
    CHECK(hipMemset(hist_d, 0, 64));
    CHECK(hipEventRecord(a));
    k_atomics<MODE><<<blocks, 256>>>(table, rows, regions, iters, hist_d);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)blocks * 256 * iters * 2;
    std::vector<float> h((size_t)rows * regions * 2);
    CHECK(hipMemcpy(h.data(), table, h.size() * 4, hipMemcpyDeviceToHost));
    double sum = 0; for (float v : h) sum += v;
    uint32_t hist[16]; CHECK(hipMemcpy(hist, hist_d, 64, hipMemcpyDeviceToHost));
    printf("%-46s %8.3f ms  %7.1f G atomics/s  %6.3f TB/s  sum %s (%.0f / %.0f)  xcd blocks:", name, ms, n / ms / 1e6,
           n * 4 / ms / 1e9, (!check || sum == n) ? "OK" : "LOST", sum, n);
    for (int i = 0; i < 8; i++) printf(" %u", hist[i]);
    printf("\n");
}

int main(int argc, char **argv)
{
    const uint32_t rows = 1u << 19;        // 4 MiB of float2 rows per region = one hashed level
    const uint32_t blocks = argc > 1 ? atoi(argv[1]) : 4096, iters = argc > 2 ? atoi(argv[2]) : 64;
    float *table; uint32_t *hist;
    CHECK(hipMalloc(&table, (size_t)rows * 16 * 8));
    CHECK(hipMalloc(&hist, 64));
    run<0>("agent scope, 16 regions (any XCD)", table, rows, 16, blocks, iters, hist, true);
    run<0>("agent scope, 1 region", table, rows, 1, blocks, iters, hist, true);
    run<3>("agent scope, region owned by XCD (8 regions)", table, rows, 8, blocks, iters, hist, true);
    run<2>("workgroup scope, region owned by XCD (8 regions)", table, rows, 8, blocks, iters, hist, true);
    run<1>("workgroup scope, 16 regions any XCD (incoherent)", table, rows, 16, blocks, iters, hist, false);
    run<1>("workgroup scope, 1 region any XCD (incoherent)", table, rows, 1, blocks, iters, hist, false);
    return 0;
}
