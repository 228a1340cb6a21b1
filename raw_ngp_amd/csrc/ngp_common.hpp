// Shared host/device helpers for libngp_hip.so (gfx950 only; no CUDA dual paths).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ngp_hip.h"

namespace ngp {

// ---------------------------------------------------------------- error plumbing
char *error_buffer();  // thread-local, 512 bytes (defined in capi_common.hip)

#define NGP_FAIL(code, ...)                                   \
    do {                                                      \
        snprintf(ngp::error_buffer(), 512, __VA_ARGS__);      \
        return (code);                                        \
    } while (0)

#define NGP_REQUIRE(cond, ...)                                \
    do {                                                      \
        if (!(cond)) NGP_FAIL(NGP_EINVAL, __VA_ARGS__);       \
    } while (0)

#define NGP_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) NGP_FAIL(NGP_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline hipStream_t as_stream(ngp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device helpers
constexpr int kWave = 64;  // CDNA wavefront

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// XCD-aware work-item remap.  Workgroups are dealt round-robin over the 8 XCDs, so the
// blocks {b : b % 8 == k} share one XCD (and its 4 MiB L2).  This bijection hands XCD
// group k the contiguous work-item range [k*n/8, (k+1)*n/8): items that share data
// (one level's hash table) then share an L2.  Placement only changes speed.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7u, k = bid & 7u, i = bid >> 3;
    const uint32_t base = k < r ? k * (q + 1) : r * (q + 1) + (k - r) * q;
    return base + i;
}

// Placement of per-level work: `n_levels` runs of `per_level` tiles.  The finest levels of a hash grid cost 5-10 x a
// coarse one (every corner is its own cache line), so handing XCD k two NEIGHBOURING levels (xcd_remap over a
// level-major item list) leaves XCD 7 with the two most expensive ones.  Snake instead: XCD k takes levels k, 15-k,
// 16+k, 31-k, ... -- a level still lives in one L2, and cheap levels are paired with expensive ones.
// Launch snake_blocks(n_levels, per_level) workgroups; level == kNoLevel means "nothing to do".
constexpr uint32_t kNoLevel = 0xffffffffu;
// (fewer than 8 levels: one level per XCD would idle the rest -- keep the contiguous split there)
__host__ __device__ inline uint32_t snake_rounds(uint32_t n_levels) { return (n_levels + 7u) / 8u; }
static inline uint32_t snake_blocks(uint32_t n_levels, uint32_t per_level)
{
    return n_levels < 8u ? n_levels * per_level : 8u * snake_rounds(n_levels) * per_level;
}
__device__ __forceinline__ void snake_level_tile(uint32_t bid, uint32_t per_level, uint32_t n_levels, uint32_t &level,
                                                 uint32_t &tile)
{
    if (n_levels < 8u) {
        const uint32_t item = xcd_remap(bid, n_levels * per_level);
        level = item / per_level;
        tile = item - level * per_level;
        return;
    }
    const uint32_t k = bid & 7u, i = bid >> 3;        // XCD, index within the XCD
    const uint32_t round = i / per_level;
    tile = i - round * per_level;
    level = (round >> 1) * 16u + ((round & 1u) ? 15u - k : k);
    if (level >= n_levels) level = kNoLevel;
}

// Placement by measured cost.  The snake balances the XCDs only when cost grows with the level the way it does for
// ray-ordered samples; for scattered points (the density-grid refresh) every hashed level costs the same 4 lines per
// point and the snake leaves three XCDs with two hashed levels and five with one.  Here a level is cut into 8 slots
// (tile % 8) and XCD k works on the slots own[k][level] of each level: the 8 * n_levels slots, level-major, are dealt
// out in 8 contiguous runs of equal cost, so a level lives in at most two or three L2s and every XCD ends together.
constexpr uint32_t kPlacedLevels = 16;
struct alignas(16) LevelPlacement {
    uint8_t own[8][kPlacedLevels];
};

static inline uint32_t placed_tiles(uint32_t mask, uint32_t per_level)
{
    return (per_level >> 3) * (uint32_t)__builtin_popcount(mask) +
           (uint32_t)__builtin_popcount(mask & ((1u << (per_level & 7u)) - 1u));
}

// cost[l] > 0: relative cost of one tile of level l.  Returns the workgroups to launch (8 x the longest XCD's list).
static inline uint32_t place_levels(LevelPlacement &p, const float *cost, uint32_t n_levels, uint32_t per_level)
{
    double total = 0.0, acc = 0.0;
    for (uint32_t l = 0; l < n_levels; l++) total += cost[l];
    for (uint32_t k = 0; k < 8; k++)
        for (uint32_t l = 0; l < kPlacedLevels; l++) p.own[k][l] = 0;
    uint32_t k = 0;
    for (uint32_t l = 0; l < n_levels; l++)
        for (uint32_t s = 0; s < 8; s++) {
            const double c = cost[l] / 8.0;
            while (k < 7 && acc + 0.5 * c > total * (k + 1) / 8.0) k++;   // the slot goes where most of it falls
            p.own[k][l] |= (uint8_t)(1u << s);
            acc += c;
        }
    uint32_t longest = 0;
    for (k = 0; k < 8; k++) {
        uint32_t n = 0;
        for (uint32_t l = 0; l < n_levels; l++) n += placed_tiles(p.own[k][l], per_level);
        longest = n > longest ? n : longest;
    }
    return 8u * longest;
}

__device__ __forceinline__ void placed_level_tile(const LevelPlacement &p, uint32_t bid, uint32_t per_level,
                                                  uint32_t n_levels, uint32_t &level, uint32_t &tile)
{
    const uint32_t k = bid & 7u;
    uint32_t i = bid >> 3;
    level = kNoLevel;
    tile = 0;
    // this XCD's row in ONE 16-byte scalar load (a load per level would be a chain of 16 latencies in front of every tile)
    static_assert(kPlacedLevels == 16, "one uint4 per XCD");
    const uint4 row = reinterpret_cast<const uint4 *>(&p.own[0][0])[k];
    const uint32_t words[4] = {row.x, row.y, row.z, row.w};
#pragma unroll
    for (uint32_t l = 0; l < kPlacedLevels; l++) {
        const uint32_t m = l < n_levels ? (words[l >> 2] >> ((l & 3u) * 8u)) & 0xffu : 0u;
        if (!m) continue;
        const uint32_t c = (uint32_t)__popc(m);
        const uint32_t n = (per_level >> 3) * c + (uint32_t)__popc(m & ((1u << (per_level & 7u)) - 1u));
        if (i < n) {
            const uint32_t q = i / c;
            uint32_t r = i - q * c, bits = m;
            while (r--) bits &= bits - 1u;            // drop the r lowest owned slots
            level = l;
            tile = q * 8u + (uint32_t)__ffs(bits) - 1u;
            return;
        }
        i -= n;
    }
}

}  // namespace ngp
