// Workspace layout and run detection of the binned table backward, shared with the slab forward kernel (which
// counts the records per chunk while it has the rows in registers anyway).
#pragma once
#include "grid_common.hpp"

namespace ngp {

constexpr uint32_t kChunkRows = 4096;    // rows per chunk: 4096 x float2 = 32 KiB of LDS
constexpr uint32_t kChunkShift = 12;
constexpr uint32_t kSeg = 32768;         // records per reduce work item
constexpr uint32_t kMaxChunks = 2048;    // LDS histogram bound of the binned path (tables up to 8 M rows)
constexpr uint32_t kSegBig = 8 * kSeg;   // ... of a heavy chunk (coarse dense levels): fewer, longer items
constexpr uint32_t kReduceBlock = 512;
constexpr int kHeadroomBits = 25;        // records that may land on one row without overflowing the int64 sum
constexpr uint32_t kFillTile = 512;      // samples per fill workgroup
constexpr uint32_t kFillBlock = 512;     // ... one per lane
constexpr uint32_t kCountTile = 2048;    // samples per count workgroup (8 per lane)

// workspace header (uint32 words); arrays sized for n_chunks_max
struct WsLayout {
    uint32_t *chunk_base;   // [kMaxLevels + 1] first chunk of each level; [L] = total chunks; [kMaxLevels + 1] = max |grad| bits
    uint32_t *count;        // [n_chunks_max]
    uint32_t *cursor;       // [n_chunks_max]
    uint32_t *offset;       // [n_chunks_max + 1] record offsets (multiples of 4)
    uint32_t *seg_base;     // [n_chunks_max + 1] first reduce work item of each chunk
    uint32_t *records;      // 3 words per record
};

__host__ __device__ inline WsLayout ws_layout(void *ws, uint32_t n_chunks_max)
{
    WsLayout w;
    uint32_t *p = reinterpret_cast<uint32_t *>(ws);
    w.chunk_base = p;
    p += kMaxLevels + 4;
    w.count = p;
    p += n_chunks_max;
    w.cursor = p;
    p += n_chunks_max;
    w.offset = p;
    p += n_chunks_max + 1;
    w.seg_base = p;
    p += n_chunks_max + 1;
    p += (4 - ((uintptr_t)(p - reinterpret_cast<uint32_t *>(ws)) & 3)) & 3;
    w.records = p;
    return w;
}

static inline size_t ws_bytes(uint32_t B, uint32_t L, uint32_t n_chunks_max)
{
    const size_t head = (size_t)(kMaxLevels + 4 + 4 * (size_t)n_chunks_max + 2 + 4) * 4;
    return head + ((size_t)B * L * 8 + 4 * (size_t)n_chunks_max + 8) * 12 + 64;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ float row_shr_f(float x)   // lane i <- lane i - N of its 16-lane row, 0 if none
{
    return __uint_as_float(dpp_u32<0x110 | N>(0u, __float_as_uint(x)));
}

constexpr uint32_t kDeadKey = 0xffffffffu;
// merging pays while consecutive samples share cells (res * step length < ~0.7); the caller passes the finest
// resolution worth merging (header word kMaxLevels + 2, written by the plan kernel; cell keys need res <= 1024)
__device__ __forceinline__ bool mergeable(const Geom<3> &g, const WsLayout &w)
{
    return g.res <= min(w.chunk_base[kMaxLevels + 2], 1024u);
}
__device__ __forceinline__ uint32_t cell_key(const Cell<3> &cl) { return cl.c[0] | (cl.c[1] << 10) | (cl.c[2] << 20); }

// must be called by all lanes of the wave.  dist = lanes back to the head of my run; returns "I am the run's tail"
__device__ __forceinline__ bool run_shape(uint32_t key, bool live, uint32_t &dist)
{
    const uint32_t l16 = threadIdx.x & 15u;
    const uint32_t prev = dpp_u32<0x111>(0xfffffffeu, key), next = dpp_u32<0x101>(0xfffffffeu, key);
    uint32_t s = (l16 == 0u || prev != key) ? l16 : 0u;
    s = max(s, dpp_u32<0x111>(0u, s));
    s = max(s, dpp_u32<0x112>(0u, s));
    s = max(s, dpp_u32<0x114>(0u, s));
    s = max(s, dpp_u32<0x118>(0u, s));
    dist = l16 - s;
    return live && (l16 == 15u || next != key);
}

}  // namespace ngp
