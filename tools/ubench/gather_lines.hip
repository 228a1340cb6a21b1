// Microbenchmark: what can the hash-grid forward's ADDRESS STREAM reach on this GPU, with the arithmetic taken away?
//
// grid_forward_slab_kernel (csrc/engine_kernels.hip) is not HBM-bound: the 46.5 MiB table lives in L2 / Infinity Cache and
// the kernel is limited by how fast single cache lines can be requested and returned.  This program issues
// exactly that kernel's loads -- same table geometry (L = 16, F = 2, T = 2^19, base 16, finest 2048: the bound-1 table of
// SURVEY.md section 8), same ray-ordered samples (rays through a shell, dt = 2 sqrt(3) / 1024 in world units), same
// level -> XCD placements, the x-neighbour pairs fetched with one 16-byte load where their rows are adjacent -- and nothing
// else: no weights, no interpolation, no slab store beyond one float2 per (sample, level) that keeps the loads alive.
// Variants: levels per thread = 1 / 2 / 4, i.e. 8 / 16 / 32 gathers in flight per lane.
// Output: microseconds per launch, "line requests" per second (4 lines per sample and hashed level, 1-2 on dense levels: the
// count the forward's TCC_REQ counter reports is printed by rocprofv3, not here) and the equivalent of the forward's
// algorithmic bytes (1164 B/sample) per second -- the ceiling `roofline_forward.frac_of_line_rate` in bench.py refers to.
//
//   hipcc -O3 --offload-arch=gfx950 -I raw_ngp_amd/csrc -o tools/bin/ubench_gather tools/ubench/gather_lines.hip
//   tools/bin/ubench_gather [samples_per_ray = 34] [rays = 4096]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <random>
#include <vector>

#include "grid_common.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using namespace ngp;

// LPT levels per thread; workgroup = 256 samples x LPT consecutive levels
// SNAKE (LPT = 1): XCD k takes levels k and 15 - k instead of 2k and 2k + 1 (ngp_common.hpp: snake_level_tile)
template <uint32_t LPT, bool PAIR, bool SNAKE = false>
__global__ __launch_bounds__(256) void gather_kernel(const float *__restrict__ x01, const float *__restrict__ table,
                                                     const int32_t *__restrict__ offsets, float2 *__restrict__ out,
                                                     uint32_t B, uint32_t nchunks, LevelRes lv)
{
    uint32_t lg, tile;
    if (SNAKE) {
        snake_level_tile(blockIdx.x, nchunks, 16u, lg, tile);
        if (lg == kNoLevel) return;
    } else {
        const uint32_t item = xcd_remap(blockIdx.x, gridDim.x);
        lg = item / nchunks;
        tile = item - lg * nchunks;
    }
    const uint32_t b = tile * 256u + threadIdx.x;
    if (b >= B) return;
    float x[3];
#pragma unroll
    for (uint32_t d = 0; d < 3; d++) x[d] = x01[(size_t)b * 3 + d];
    float2 v[LPT][8];
#pragma unroll
    for (uint32_t k = 0; k < LPT; k++) {
        const uint32_t level = lg * LPT + k;
        const Geom<3> g = make_geom<3>(offsets, level, lv.res[level], 0u);
        const float *__restrict__ tab = table + (size_t)(uint32_t)offsets[level] * 2;
        Cell<3> cl = {};
        locate<3>(x, g.res, false, 0u, cl);
        const AxisTerms<3> terms = axis_terms<3>(g, cl);
#pragma unroll
        for (uint32_t yz = 0; yz < 4; yz++) {
            const uint32_t ra = row_from_terms<3>(g, terms, yz * 2u), rb = row_from_terms<3>(g, terms, yz * 2u + 1u);
            if (PAIR && (rb == ra + 1u || ra == rb + 1u)) {
                const float4 t = *reinterpret_cast<const float4 *>(tab + (size_t)min(ra, rb) * 2);
                v[k][yz * 2] = make_float2(t.x, t.y);
                v[k][yz * 2 + 1] = make_float2(t.z, t.w);
            } else {
                v[k][yz * 2] = *reinterpret_cast<const float2 *>(tab + (size_t)ra * 2);
                v[k][yz * 2 + 1] = *reinterpret_cast<const float2 *>(tab + (size_t)rb * 2);
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < LPT; k++) {
        float2 s = make_float2(0.f, 0.f);
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) {
            s.x += v[k][c].x;
            s.y += v[k][c].y;
        }
        out[(size_t)(lg * LPT + k) * B + b] = s;
    }
}

int main(int argc, char **argv)
{
    const uint32_t K = argc > 1 ? atoi(argv[1]) : 34, N = argc > 2 ? atoi(argv[2]) : 4096;
    const uint32_t B = N * K, L = 16, H = 16;
    // the level table of gridencoder/grid.py:102-146 for desired_resolution 2048 (bound 1), T = 2^19
    const double scale = exp2(log2(2048.0 / H) / (L - 1));
    const float S = (float)log2(scale);
    std::vector<int32_t> offsets(L + 1, 0);
    for (uint32_t l = 0; l < L; l++) {
        const uint64_t res = (uint64_t)ceil(H * pow(scale, (double)l));
        uint64_t rows = std::min<uint64_t>(1ull << 19, res * res * res);
        rows = (rows + 7) / 8 * 8;
        offsets[l + 1] = offsets[l] + (int32_t)rows;
    }
    LevelRes lv;
    fill_levels(lv, S, H, L);
    // ray-ordered samples: rays aimed at points inside a ball of radius 0.4, marched through a shell in front of them
    std::mt19937 rng(0);
    std::normal_distribution<float> nd;
    std::uniform_real_distribution<float> ud(-0.4f, 0.4f);
    std::vector<float> x(3 * (size_t)B);
    const float dt = 2.0f * sqrtf(3.0f) / 1024.0f;
    for (uint32_t r = 0; r < N; r++) {
        float o[3], t[3], d[3], no = 0, nd_ = 0;
        for (int a = 0; a < 3; a++) { o[a] = nd(rng); no += o[a] * o[a]; t[a] = ud(rng); }
        for (int a = 0; a < 3; a++) { o[a] *= 3.0f / sqrtf(no); d[a] = t[a] - o[a]; nd_ += d[a] * d[a]; }
        const float len = sqrtf(nd_);
        for (int a = 0; a < 3; a++) d[a] /= len;
        for (uint32_t k = 0; k < K; k++) {
            const float tt = len - 0.06f + dt * k;
            for (int a = 0; a < 3; a++) {
                const float w = std::min(0.999f, std::max(-0.999f, o[a] + d[a] * tt));
                x[((size_t)r * K + k) * 3 + a] = (w + 1.0f) * 0.5f;
            }
        }
    }
    float *d_x, *d_tab;
    int32_t *d_off;
    float2 *d_out;
    const size_t rows = offsets[L];
    CHECK(hipMalloc(&d_x, x.size() * 4));
    CHECK(hipMalloc(&d_tab, rows * 8));
    CHECK(hipMalloc(&d_off, (L + 1) * 4));
    CHECK(hipMalloc(&d_out, (size_t)L * B * 8));
    CHECK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_tab, 0, rows * 8));
    CHECK(hipMemcpy(d_off, offsets.data(), (L + 1) * 4, hipMemcpyHostToDevice));
    const uint32_t nchunks = (B + 255) / 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%u samples (%u rays x %u), table %zu rows = %.1f MiB, levels 0-4 dense\n", B, N, K, rows, rows * 8 / 1048576.0);
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        CHECK(hipDeviceSynchronize());
        const int iters = 50;
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < iters; i++) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        const double lines = (double)B * (11 * 4 + 5 * 1.5);   // nominal: 4 lines per hashed level, ~1.5 on the dense ones
        printf("%-34s %8.1f us   %6.1f G lines/s (nominal)   %5.2f TB/s of the forward's 1164 B/sample = %.3f of 8 TB/s\n", name,
               us, lines / us / 1e3, (double)B * 1164 / us / 1e6, (double)B * 1164 / us / 1e6 / 8.0);
    };
    run("8 gathers in flight, 8-byte loads", [&] {
        gather_kernel<1, false><<<nchunks * 16, 256>>>(d_x, d_tab, d_off, d_out, B, nchunks, lv);
    });
    run("8 in flight, paired 16-byte loads", [&] {
        gather_kernel<1, true><<<nchunks * 16, 256>>>(d_x, d_tab, d_off, d_out, B, nchunks, lv);
    });
    run("8 in flight, 8-byte, snake placement", [&] {
        gather_kernel<1, false, true><<<snake_blocks(16, nchunks), 256>>>(d_x, d_tab, d_off, d_out, B, nchunks, lv);
    });
    run("8 in flight, paired, snake placement", [&] {
        gather_kernel<1, true, true><<<snake_blocks(16, nchunks), 256>>>(d_x, d_tab, d_off, d_out, B, nchunks, lv);
    });
    run("16 in flight, paired", [&] {
        gather_kernel<2, true><<<nchunks * 8, 256>>>(d_x, d_tab, d_off, d_out, B, nchunks, lv);
    });
    run("32 in flight, paired", [&] {
        gather_kernel<4, true><<<nchunks * 4, 256>>>(d_x, d_tab, d_off, d_out, B, nchunks, lv);
    });
    return 0;
}
