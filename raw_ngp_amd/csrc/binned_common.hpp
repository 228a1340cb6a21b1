// Workspace layout and run detection of the binned table backward, shared with the slab forward kernel (which
// counts the records per chunk while it has the rows in registers anyway).
#pragma once
#include "grid_common.hpp"

namespace ngp {

// rows per chunk.  The reduce workgroup of a chunk holds its sums as 2 x int64 per row in LDS: 4096 rows = 64 KiB, two
// workgroups per CU.  -DNGP_CHUNK_SHIFT=11 builds the 2048-row variant (32 KiB, three workgroups per CU at 80 registers a
// lane, optimiser state requested after the record gather): measured even with this one in the step (0.3244 / 0.3318 against
// 0.3270 / 0.3301 ms, same-box pairs) although ONE workgroup per CU costs 55 % (124 against 80 us) -- beyond two the kernel
// is not short of workgroups in flight; twice the directory words and quad padding eat what the third one brings.
#ifndef NGP_CHUNK_SHIFT
#define NGP_CHUNK_SHIFT 12
#endif
constexpr uint32_t kChunkShift = NGP_CHUNK_SHIFT;
constexpr uint32_t kChunkRows = 1u << kChunkShift;
constexpr uint32_t kSeg = 32768;         // records per reduce work item
constexpr uint32_t kMaxChunks = (1u << 23) >> kChunkShift;   // LDS histogram bound of the binned path (tables up to 8 M rows)
constexpr uint32_t kSegBig = 8 * kSeg;   // ... of a heavy chunk (coarse dense levels): fewer, longer items
constexpr uint32_t kReduceBlock = 512;
constexpr int kHeadroomBits = 25;        // records that may land on one row without overflowing the int64 sum
#ifndef NGP_FILL_TILE
#define NGP_FILL_TILE 512
#endif
constexpr uint32_t kFillTile = NGP_FILL_TILE;   // samples per fill workgroup (a multiple of 256; -DNGP_FILL_TILE=256 to try)
constexpr uint32_t kFillBlock = kFillTile;      // ... one per lane
constexpr uint32_t kFillTailPer = kFillBlock / 256u;   // passenger groups (256 lanes each) per fill workgroup
constexpr uint32_t kCountTile = 2048;    // samples per count workgroup (8 per lane)

// workspace header (uint32 words); arrays sized for n_chunks_max
struct WsLayout {
    uint32_t *chunk_base;   // [kMaxLevels + 1] first chunk of each level; [L] = total chunks; [kMaxLevels + 1] = max |grad| bits
    uint32_t *count;        // [n_chunks_max]
    uint32_t *cursor;       // [n_chunks_max]
    uint32_t *offset;       // [n_chunks_max + 1] record offsets (multiples of 4)
    uint32_t *seg_base;     // [n_chunks_max + 1] first reduce work item of each chunk
    float2 *vals;           // record payloads (w * g.x, w * g.y), rec_cap of them
    uint16_t *keys;         // record keys (row inside the chunk), rec_cap of them, behind the payloads
};

// records a call can emit at most: 8 per (sample, level) + the 4-record alignment slack of every chunk
__host__ __device__ inline size_t ws_rec_cap(uint32_t B, uint32_t L, uint32_t n_chunks_max)
{
    return (size_t)B * L * 8 + 4 * (size_t)n_chunks_max + 8;
}

// rec_cap = 0: header only (callers that never touch the records: counting, scanning)
__host__ __device__ inline WsLayout ws_layout(void *ws, uint32_t n_chunks_max, size_t rec_cap = 0)
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
    w.vals = reinterpret_cast<float2 *>(p);
    w.keys = reinterpret_cast<uint16_t *>(w.vals + rec_cap);
    return w;
}

// ---- tile-local layout of the records (bin_fill_local_kernel / bin_reduce_local_kernel) --------------------------------
// A fill workgroup (kFillTile samples of one level) owns a fixed region of kRegion record slots, writes its records
// there sorted by chunk -- every chunk's run starts at a multiple of four slots and is padded to one with null records (key 0,
// value 0), so that the reduce reads whole quads with 16-byte loads -- and leaves one directory word per chunk:
// dir[chunk][tile] = first slot | count << 16.  Nothing is
// coordinated across workgroups (no cursors, no counting pass, no scan); the reduce workgroup of a chunk collects its runs
// from every tile through the directory column of that chunk.
constexpr uint32_t kMaxTiles = 2048;     // fill tiles per call the tile-local reduce can index (B <= 1 Mi samples)
__host__ __device__ inline uint32_t ws_tiles(uint32_t B) { return (B + kFillTile - 1) / kFillTile; }
constexpr uint32_t kLocalBins = (1u << 19) >> kChunkShift;    // chunks per level the tile-local layout is sized for (T = 2^19)
constexpr uint32_t kRegion = kFillTile * 8 + 4 * kLocalBins;  // 8 records per sample + up to 3 slots of padding per chunk
__host__ __device__ inline size_t ws_rec_cap_local(uint32_t B, uint32_t L) { return (size_t)ws_tiles(B) * L * kRegion; }
// the directory sits behind the key stream (4-byte aligned: the key stream has an even number of entries)
__host__ __device__ inline uint32_t *ws_dir(const WsLayout &w, size_t rec_cap)
{
    return reinterpret_cast<uint32_t *>(const_cast<uint16_t *>(w.keys) + ((rec_cap + 1) & ~(size_t)1));
}

static inline size_t ws_bytes(uint32_t B, uint32_t L, uint32_t n_chunks_max)
{
    const size_t head = (size_t)(kMaxLevels + 4 + 4 * (size_t)n_chunks_max + 2 + 4) * 4;
    const size_t rec = ws_rec_cap(B, L, n_chunks_max) > ws_rec_cap_local(B, L) ? ws_rec_cap(B, L, n_chunks_max)
                                                                               : ws_rec_cap_local(B, L);
    // (10 bytes per record are used: float2 + uint16; the slack holds the tile-local layout's directory)
    const size_t dir = (size_t)n_chunks_max * ws_tiles(B) * 4;
    const size_t need_local = ws_rec_cap_local(B, L) * 10 + dir + 16;
    const size_t need = rec * 12 > need_local ? rec * 12 : need_local;
    return head + need + 64;
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

// exclusive prefix of the per-chunk record counts (4-record aligned) -> record offsets, plus the reduce work items; to be
// run by ONE workgroup of 256 .. 1024 lanes (bin_scan_kernel, or the fused step's step_begin work)
__device__ __forceinline__ void bin_scan_block(uint32_t L, const WsLayout &w, bool single_segment)
{
    __shared__ uint32_t wave_a[16], wave_b[16];
    __shared__ uint32_t carry_a, carry_b;
    const uint32_t n = w.chunk_base[L];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6, width = blockDim.x, nw = width >> 6;
    if (tid == 0) carry_a = carry_b = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += width) {
        const uint32_t i = base + tid;
        const uint32_t cnt = i < n ? w.count[i] : 0u;
        const uint32_t cnt4 = (cnt + 3u) & ~3u;   // 4-record alignment: 16-byte loads in the reduce kernel
        const uint32_t seg_len = cnt > kSegBig ? kSegBig : kSeg;   // heavy chunks: 8x longer work items
        const uint32_t seg = i < n ? (single_segment ? 1u : max(1u, (cnt + seg_len - 1) / seg_len)) : 0u;
        uint32_t a = cnt4, s = seg;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t ua = __shfl_up(a, d, 64), us = __shfl_up(s, d, 64);
            if (lane >= d) {
                a += ua;
                s += us;
            }
        }
        if (lane == 63u) {
            wave_a[wid] = a;
            wave_b[wid] = s;
        }
        __syncthreads();
        uint32_t oa = 0, ob = 0, ta = 0, tb = 0;
        for (uint32_t k = 0; k < nw; k++) {
            if (k < wid) {
                oa += wave_a[k];
                ob += wave_b[k];
            }
            ta += wave_a[k];
            tb += wave_b[k];
        }
        const uint32_t ca = carry_a, cb = carry_b;
        if (i < n) {
            w.offset[i] = ca + oa + a - cnt4;
            w.seg_base[i] = cb + ob + s - seg;
        }
        __syncthreads();
        if (tid == 0) {
            carry_a = ca + ta;
            carry_b = cb + tb;
        }
        __syncthreads();
    }
    if (tid == 0) {
        w.offset[n] = carry_a;
        w.seg_base[n] = carry_b;
    }
}

// ------------------------------------------------------------------ per-step scalars (+ the scan above)
// What the fused step does once per step with one workgroup: learning rate and Adam bias corrections (the LambdaLR of
// main.py:261 and torch.optim.Adam's), loss = 0, samples_seen += this batch's sample count, step counter + 1 -- and, if
// asked, the record offsets of the binned table backward.  A kernel of its own (ngp_x_step_begin), or one extra workgroup
// of the MLP forward's launch (ngp_x_mlp_forward_step_begin): nothing consumes any of it before the compositor.
struct StepBegin {
    uint32_t *step_counter;
    float *hyper;
    double lr0, decay_steps, b1, b2;
    float *loss_out;
    long long *samples_seen;
    const int32_t *sample_counter;
    bool scan, single_segment;
    uint32_t L;
    WsLayout w;
    // dynamic loss scale (mlp_common.hpp: LossScalerWord; NULL: none): the previous step is settled here -- GradScaler.update()
    float *scaler;
    float growth, backoff;
    uint32_t growth_interval;
};
__device__ __forceinline__ void step_begin_block(const StepBegin &a)
{
    // the three scalars cost a double-precision pow each (~1 us on a lone lane): three different waves take one each
    // while the others already wait on the scan's first loads
    const uint32_t done = a.step_counter[0];
    const uint32_t nw = blockDim.x >> 6, wid = threadIdx.x >> 6;
    // Dynamic loss scale: what torch.cuda.amp.GradScaler.update() does after a step (train_utils.py:897-904), done at the
    // start of the next one, where a single workgroup runs anyway.  A step that saw a non-finite gradient was skipped by
    // every optimiser kernel: the scale backs off, Adam's t does not advance (its bias corrections below follow the
    // optimiser steps TAKEN, as torch's do); the learning-rate schedule follows the step counter either way
    // (lr_scheduler.step() is unconditional, train_utils.py:906-907).
    uint32_t *sw = reinterpret_cast<uint32_t *>(a.scaler);
    uint32_t found = 0, pending = 0, taken = done;
    if (sw) {
        found = sw[2], pending = sw[6];
        taken = sw[4] + ((pending && !found) ? 1u : 0u);
    }
    if ((threadIdx.x & 63u) == 0) {
        const double t = (double)taken + 1.0;
        if (wid == nw - 1) a.hyper[0] = (float)(a.lr0 * pow(0.1, fmin((double)done / a.decay_steps, 1.0)));
        if (wid == (nw >= 2 ? nw - 2 : 0)) a.hyper[1] = (float)(1.0 - pow(a.b1, t));
        if (wid == (nw >= 3 ? nw - 3 : 0)) a.hyper[2] = (float)(1.0 / sqrt(1.0 - pow(a.b2, t)));
    }
    if (a.scan) bin_scan_block(a.L, a.w, a.single_segment);
    __syncthreads();   // every wave has read the step counter (and the scaler's words)
    if (threadIdx.x != 0) return;
    a.step_counter[0] = done + 1u;
    if (sw) {
        float scale = a.scaler[0];
        uint32_t tracker = sw[3];
        if (pending) {
            if (found) {
                scale *= a.backoff;
                tracker = 0;
                sw[5] += 1u;
            } else if (++tracker >= a.growth_interval) {
                scale = fminf(scale * a.growth, 0x1p+60f);
                tracker = 0;
            }
        }
        a.scaler[0] = scale;
        a.scaler[1] = 1.0f / scale;
        sw[2] = 0u;
        sw[3] = tracker;
        sw[4] = taken;
        sw[6] = 1u;
    }
    if (a.loss_out) a.loss_out[0] = 0.0f;
    if (a.samples_seen && a.sample_counter) a.samples_seen[0] += (long long)a.sample_counter[0];
}
// host side: checks and packing shared by the two entry points
int step_begin_args(StepBegin &a, const char *who, uint32_t *step_counter, float *hyper, double lr0, double decay_steps,
                    double beta1, double beta2, float *loss_out, int64_t *samples_seen, const int32_t *sample_counter,
                    void *binned_workspace, uint32_t L, uint32_t n_rows_total, int single_segment, float *scaler = nullptr,
                    double growth = 2.0, double backoff = 0.5, uint32_t growth_interval = 2000);

}  // namespace ngp
