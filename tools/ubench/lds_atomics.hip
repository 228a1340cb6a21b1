// Microbenchmark: throughput of LDS atomic adds by type (f32 / u32 / u64) and of an explicit
// read-add-write on random addresses, 256-thread workgroups, 8192-dword image.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, uint32_t iters)
{
    __shared__ unsigned long long img64[4096];
    float *imgf = reinterpret_cast<float *>(img64);
    uint32_t *imgu = reinterpret_cast<uint32_t *>(img64);
    for (uint32_t i = threadIdx.x; i < 4096; i += 256) img64[i] = 0;
    __syncthreads();
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    for (uint32_t i = 0; i < iters; i++) {
        const uint32_t r = hash32(gid * 977u + i * 0x9e3779b9u) & 4095u;
        if (MODE == 0) { atomicAdd(&imgf[r * 2], 1.0f); atomicAdd(&imgf[r * 2 + 1], 2.0f); }
        if (MODE == 1) { atomicAdd(&imgu[r * 2], 1u); atomicAdd(&imgu[r * 2 + 1], 2u); }
        if (MODE == 2) { atomicAdd(&img64[r], 3ull); }
        if (MODE == 3) {   // non-atomic read-modify-write of the float2 row (wrong under collisions; rate only)
            float2 v = reinterpret_cast<float2 *>(imgf)[r];
            v.x += 1.0f; v.y += 2.0f;
            reinterpret_cast<float2 *>(imgf)[r] = v;
        }
    }
    __syncthreads();
    float s = 0;
    for (uint32_t i = threadIdx.x; i < 8192; i += 256) s += imgf[i];
    if (s == 123.456f) out[gid] = s;
}

template <int MODE>
static void run(const char *name, float *out, uint32_t blocks, uint32_t iters)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<MODE><<<blocks, 256>>>(out, 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    k<MODE><<<blocks, 256>>>(out, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double rows = (double)blocks * 256 * iters;
    printf("%-34s %8.3f ms   %8.1f G row-updates/s (chip)   %.2f row-updates/clk/CU @2.1GHz\n", name, ms, rows / ms / 1e6,
           rows / ms / 1e6 / 256 / 2.1);
}

int main()
{
    float *out; CHECK(hipMalloc(&out, 1 << 26));
    const uint32_t blocks = 256 * 4, iters = 2048;
    run<0>("ds_add_f32 x2 (float2 row)", out, blocks, iters);
    run<1>("ds_add_u32 x2", out, blocks, iters);
    run<2>("ds_add_u64 x1", out, blocks, iters);
    run<3>("ds_read_b64 + add + ds_write_b64", out, blocks, iters);
    return 0;
}
