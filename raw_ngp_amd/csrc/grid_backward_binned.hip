// Hash-grid table gradient without scattered global float atomics (D = 3, C = 2: the field's encoder).
//
// Why: the reference's kernel_grid_backward (gridencoder/src/gridencoder.cu:252-349) issues one float
// atomicAdd per (sample, level, corner, channel).  On MI355X global float atomics execute memory-side
// at ~21 G requests/s when lanes hit different 64-B lines (measured: tools/ubench/atomics.hip), i.e.
// 3.3 ms for 2^18 samples -- 8x the whole step budget.  Plain stores and LDS atomics are ~50x faster,
// (and LDS *float* atomics turn out to be slow too: 0.19 row updates/clk/CU against 5.1 for ds_add_u64,
// tools/ubench/lds_atomics.hip), so the scatter becomes "bin, then reduce each bin in LDS in 64-bit fixed point":
//
//   count    chunk = 4096 consecutive rows of one level (32 KiB of float2 accumulators).  Every (sample,
//            level) of the remaining levels computes its 8 corner rows; a per-workgroup LDS histogram over
//            the level's chunks is added to the global per-chunk counts
//   scan     exclusive prefix of the counts (4-record aligned) -> record offsets; chunks heavier than kSeg
//            records are split into segments
//   fill     same traversal; records {row-in-chunk, w*g.x, w*g.y} (12 B) are first sorted by chunk inside
//            the workgroup (LDS), then written so that consecutive lanes store consecutive records
//   reduce   one workgroup per (chunk, segment): stream the records (16-byte loads, 8 records in flight per
//            lane), convert each value to 64-bit fixed point (scale 2^k, k from the largest |gradient| of the
//            call so that 2^25 records per row cannot overflow) and ds_add_u64 into a 64 KiB LDS image of the
//            chunk; integer sums are exact and order-independent.  Then grad_table[chunk] += image (plain
//            read-modify-write when the chunk has one segment, contiguous float atomics otherwise)
//
// Result: the reference's sums, each rounded once (resolution 2^-37 of the largest contribution) instead of
// once per atomic in hardware order; bitwise reproducible except for the few multi-segment chunks.
#include <type_traits>

#include "binned_common.hpp"
#include "mlp_common.hpp"

namespace ngp {

// ------------------------------------------------------------------ plan
__global__ __launch_bounds__(1024) void bin_plan_kernel(const int32_t *__restrict__ offsets, uint32_t L,
                                                        uint32_t n_chunks_max, uint32_t merge_max_res, WsLayout w)
{
    __shared__ uint32_t total;
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t l = 0; l < L; l++) {
            w.chunk_base[l] = run;
            const uint32_t T = (uint32_t)(offsets[l + 1] - offsets[l]);
            run += (T + kChunkRows - 1) >> kChunkShift;
        }
        w.chunk_base[L] = run;
        w.chunk_base[kMaxLevels + 1] = 0;   // max |grad| of this call, as float bits
        w.chunk_base[kMaxLevels + 2] = merge_max_res ? merge_max_res : 1024u;
        total = run;
    }
    __syncthreads();
    const uint32_t n = min(total, n_chunks_max);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        w.count[i] = 0;
        w.cursor[i] = 0;
    }
}

// ------------------------------------------------------------------ shared traversal piece
// rows of the 8 corners of sample b at one level; false when the sample is outside [0,1]^3
// in_bound > 0: `inputs` are world positions in [-in_bound, in_bound], mapped like the encoder does (grid.py:161)
__device__ __forceinline__ bool corner_rows(const float *__restrict__ inputs, uint32_t b, const Geom<3> &g,
                                            bool align_corners, uint32_t interp, Cell<3> &cl, uint32_t (&rows)[8],
                                            float in_bound = 0.0f)
{
    float x[3];
#pragma unroll
    for (uint32_t d = 0; d < 3; d++) {
        x[d] = inputs[(size_t)b * 3 + d];
        if (in_bound > 0.0f) x[d] = (x[d] + in_bound) / (2.0f * in_bound);
    }
    if (!locate<3>(x, g.res, align_corners, interp, cl)) return false;
    const AxisTerms<3> a = axis_terms<3>(g, cl);
#pragma unroll
    for (uint32_t corner = 0; corner < 8; corner++) rows[corner] = row_from_terms<3>(g, a, corner);
    return true;
}

// ------------------------------------------------------------------ run merging
// Samples are ray-ordered, so up to level ~10 consecutive samples fall into the same cell and would emit records
// for the same 8 rows.  Inside each row of 16 lanes (DPP reach: no LDS traffic) consecutive lanes with the same
// cell form a run; the run's contributions are summed with a segmented scan and only its last lane emits records.
// At 4096 rays x ~70 samples this halves the record count and removes the same-address pile-ups of the coarse levels
// in the LDS histogram and in the reduce kernel's ds_add_u64.  Count and fill use the same lane <-> sample mapping
// (tiles start at multiples of 16), hence see the same runs.
// segmented inclusive sum along the run: step d adds the value d lanes back when that lane belongs to my run.
// Written as fma(shifted, flag_d, v) with flag_d in {0, 1} so that the shift and the add fuse into one
// v_fmac_f32_dpp (16 values x 4 steps = 64 instructions per lane instead of 192)
struct RunFlags {
    float f1, f2, f4, f8;
};
__device__ __forceinline__ RunFlags run_flags(uint32_t dist)
{
    return {dist >= 1u ? 1.0f : 0.0f, dist >= 2u ? 1.0f : 0.0f, dist >= 4u ? 1.0f : 0.0f, dist >= 8u ? 1.0f : 0.0f};
}
__device__ __forceinline__ float run_sum(float v, const RunFlags &f)
{
#if NGP_RUNSUM_SELECT
    float t;
    t = row_shr_f<1>(v); v += f.f1 != 0.0f ? t : 0.0f;
    t = row_shr_f<2>(v); v += f.f2 != 0.0f ? t : 0.0f;
    t = row_shr_f<4>(v); v += f.f4 != 0.0f ? t : 0.0f;
    t = row_shr_f<8>(v); v += f.f8 != 0.0f ? t : 0.0f;
#else
    v = fmaf(row_shr_f<1>(v), f.f1, v);
    v = fmaf(row_shr_f<2>(v), f.f2, v);
    v = fmaf(row_shr_f<4>(v), f.f4, v);
    v = fmaf(row_shr_f<8>(v), f.f8, v);
#endif
    return v;
}

// ------------------------------------------------------------------ count
__global__ __launch_bounds__(kBlock) void bin_count_kernel(const float *__restrict__ inputs,
                                                          const int32_t *__restrict__ offsets,
                                                          const int32_t *__restrict__ B_dev, uint32_t B_cap,
                                                          uint32_t ntiles, LevelRes lv, uint32_t gridtype,
                                                          bool align_corners, uint32_t interp, float in_bound, WsLayout w)
{
    extern __shared__ uint32_t hist[];
    const uint32_t B = B_dev ? min((uint32_t)max(B_dev[0], 0), B_cap) : B_cap;
    const uint32_t item = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t level = item / ntiles;
    const uint32_t b0 = (item - level * ntiles) * kCountTile;
    if (b0 >= B) return;
    const Geom<3> g = make_geom<3>(offsets, level, lv.res[level], gridtype);
    const uint32_t first = w.chunk_base[level];
    const uint32_t nbins = w.chunk_base[level + 1] - first;
    for (uint32_t i = threadIdx.x; i < nbins; i += kBlock) hist[i] = 0;
    __syncthreads();
    const bool merge = mergeable(g, w);
#pragma unroll 2
    for (uint32_t k = 0; k < kCountTile / kBlock; k++) {
        const uint32_t b = b0 + k * kBlock + threadIdx.x;
        if (b0 + k * kBlock >= B) break;                 // uniform
        Cell<3> cl;
        uint32_t rows[8];
        const bool live = b < B && corner_rows(inputs, b, g, align_corners, interp, cl, rows, in_bound);
        bool emit = live;
        if (merge) {
            uint32_t dist;
            emit = run_shape(live ? cell_key(cl) : kDeadKey, live, dist);
        }
        if (!emit) continue;
#pragma unroll
        for (uint32_t corner = 0; corner < 8; corner++) atomicAdd(&hist[rows[corner] >> kChunkShift], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nbins; i += kBlock)
        if (hist[i]) atomicAdd(&w.count[first + i], hist[i]);
}

// ------------------------------------------------------------------ scan
__global__ __launch_bounds__(1024) void bin_scan_kernel(uint32_t L, WsLayout w, bool single_segment)
{
    bin_scan_block(L, w, single_segment);
}

// Diagnostic build only (-DNGP_STAMP_FILL, never shipped; tools/fill_stamps.py): s_memtime at the phase boundaries of the
// fill kernel as seen by thread 0 (wave 0, the wave that scans), summed per level.  No stamp executes in the product build.
#ifdef NGP_STAMP_FILL
__device__ unsigned long long ngp_dbg_fill_stamps[kMaxLevels][8];
#define NGP_FILL_STAMP_DECL unsigned long long st_prev = 0
#define NGP_FILL_STAMP(i)                                                                                  \
    do {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        unsigned long long st_now;                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory"); \
        if (threadIdx.x == 0 && (i) > 0) atomicAdd(&ngp_dbg_fill_stamps[level][(i)], st_now - st_prev);    \
        if (threadIdx.x == 0 && (i) == 0) atomicAdd(&ngp_dbg_fill_stamps[level][0], 1ull);                 \
        st_prev = st_now;                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    } while (0)
#else
#define NGP_FILL_STAMP_DECL
#define NGP_FILL_STAMP(i)
#endif

// ------------------------------------------------------------------ fill
// One workgroup = 512 samples of one level.  Records (row-in-chunk, w * g.x, w * g.y) are sorted by chunk inside the
// workgroup (LDS) and leave it as two contiguous streams per chunk: float2 payloads and uint16 keys (10 bytes a record).
// LDS: hist[nbins] | lbase[nbins] | delta[nbins] | stage_key[8 * kFillTile] u32 | stage_val[8 * kFillTile] float2
// Phases (4 barriers): zero hist | locate, weights, run merge, hist atomics | wave 0: scan -> lbase, cursor atomics in
// flight | stage (wave 0 first turns the returned cursors into delta = global slot - staging slot) | stream out
__global__ __launch_bounds__(kFillBlock) void bin_fill_kernel(const float *__restrict__ grad, const float *__restrict__ inputs,
                                                         const int32_t *__restrict__ offsets,
                                                         const int32_t *__restrict__ B_dev, uint32_t B_cap, uint32_t gstride,
                                                         uint32_t ntiles, uint32_t nbins_cap, LevelRes lv,
                                                         uint32_t gridtype, bool align_corners, uint32_t interp,
                                                         WsLayout w, uint32_t n_tail, MlpDwReduce tail)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // The first n_tail workgroups are passengers: they reduce the tiny MLPs' weight gradients (two groups of 64 outputs
    // each, mlp_common.hpp) instead of filling records.  That reduction is 10 us of latency-bound loads which nothing
    // waits for until the next step's MLP forward; as a kernel of its own it sat on the step's critical path with the
    // gap of a dependent launch on top.
    if (blockIdx.x < n_tail) {
        float(*part)[64] = reinterpret_cast<float(*)[64]>(lds) + 4 * (threadIdx.x >> 8);
        mlp_reduce_dw_group(tail, blockIdx.x * 2u + (threadIdx.x >> 8), threadIdx.x & 255u, part);
        return;
    }
    uint32_t *hist = lds, *lbase = lds + nbins_cap, *delta = lds + 2 * nbins_cap;
    uint32_t *stage_key = lds + 3 * nbins_cap;
    float2 *stage_val = reinterpret_cast<float2 *>(stage_key + 8 * kFillTile);   // 8-byte aligned: nbins_cap % 4 == 0
    __shared__ uint32_t s_total;

    const uint32_t B = B_dev ? min((uint32_t)max(B_dev[0], 0), B_cap) : B_cap;
    const uint32_t item = xcd_remap(blockIdx.x - n_tail, gridDim.x - n_tail);
    const uint32_t level = item / ntiles;
    const uint32_t b0 = (item - level * ntiles) * kFillTile;
    if (b0 >= B) return;
    NGP_FILL_STAMP_DECL;
    NGP_FILL_STAMP(0);
    const Geom<3> g = make_geom<3>(offsets, level, lv.res[level], gridtype);
    const uint32_t first = w.chunk_base[level];
    const uint32_t nbins = w.chunk_base[level + 1] - first;
    for (uint32_t i = threadIdx.x; i < nbins; i += kFillBlock) hist[i] = 0;
    __syncthreads();
    NGP_FILL_STAMP(1);   // header words + histogram zeroed

    static_assert(kFillTile == kFillBlock, "one sample per lane");
    const uint32_t b = b0 + threadIdx.x;
    uint32_t rows[8], pos[8];
    float vx[8], vy[8];
    Cell<3> cl = {};
    bool live = b < B;
    if (live) {
        float x[3];
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) x[d] = inputs[(size_t)b * 3 + d];
        live = locate<3>(x, g.res, align_corners, interp, cl);
    }
    float2 gr = make_float2(0.0f, 0.0f);
    if (live) gr = reinterpret_cast<const float2 *>(grad)[(size_t)level * gstride + b];
    NGP_FILL_STAMP(2);   // position and gradient have arrived
    const bool nan = !(gr.x == gr.x && gr.y == gr.y);
    // a sample whose gradient is exactly zero at this level (behind the compositor's early stop: a third of the samples
    // late in training) contributes nothing: no weights, and no records unless its run has a non-zero member
    const bool active = live && (gr.x != 0.0f || gr.y != 0.0f);
#pragma unroll
    for (uint32_t corner = 0; corner < 8; corner++) vx[corner] = vy[corner] = 0.0f;
    if (active) {
#pragma unroll
        for (uint32_t corner = 0; corner < 8; corner++) {
            float wgt = 1.0f;
#pragma unroll
            for (uint32_t d = 0; d < 3; d++) wgt *= (corner & (1u << d)) ? cl.f[d] : 1.0f - cl.f[d];
            vx[corner] = wgt * gr.x;
            vy[corner] = wgt * gr.y;
        }
    }
    // runs are shaped exactly as when the records were counted (every in-range sample), so a bin can only receive
    // fewer records than were reserved for it, never more
    bool emit = active;
    if (mergeable(g, w)) {
        uint32_t dist;
        const bool tail = run_shape(live ? cell_key(cl) : kDeadKey, live, dist);
        const RunFlags flags = run_flags(dist);
        // (evaluated by every lane before the &&: a DPP scan under a divergent EXEC mask reads disabled lanes)
        const float active_in_run = run_sum(active ? 1.0f : 0.0f, flags);
        emit = tail && active_in_run > 0.0f;
        // (a wave without a single active sample has nothing to add up: wave-uniform, so the scans below stay legal)
        if (__ballot(active) != 0ull) {
#pragma unroll
            for (uint32_t corner = 0; corner < 8; corner++) {
                vx[corner] = run_sum(vx[corner], flags);
                vy[corner] = run_sum(vy[corner], flags);
            }
        }
    }
    if (emit) {   // only a run's last lane needs the rows (hashes)
        const AxisTerms<3> terms = axis_terms<3>(g, cl);
#pragma unroll
        for (uint32_t corner = 0; corner < 8; corner++) {
            rows[corner] = row_from_terms<3>(g, terms, corner);
            pos[corner] = atomicAdd(&hist[rows[corner] >> kChunkShift], 1u);
        }
    }
    NGP_FILL_STAMP(3);   // weights, run sums, hashes, histogram atomics (wave 0)
    __syncthreads();
    NGP_FILL_STAMP(4);   // ... until the last wave is there

    // wave 0: exclusive scan of hist -> staging offsets (lbase); reserve the global ranges with one atomic per non-empty
    // bin -- issued here, consumed after the next barrier, so their round trip overlaps the other waves' staging
    const uint32_t lane = threadIdx.x & 63u;
    const bool fast = nbins <= 128u;               // two bins per lane of wave 0 (every level of the field's table)
    uint32_t g0 = 0, g1 = 0, n0 = 0, n1 = 0, run0 = 0;
    if (threadIdx.x < 64u) {
        if (fast) {
            const uint32_t i0 = 2u * lane, i1 = i0 + 1u;
            n0 = i0 < nbins ? hist[i0] : 0u;
            n1 = i1 < nbins ? hist[i1] : 0u;
            uint32_t inc = n0 + n1;
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                const uint32_t up = __shfl_up(inc, d, 64);
                if (lane >= d) inc += up;
            }
            run0 = inc - (n0 + n1);
            if (i0 < nbins) lbase[i0] = run0;
            if (i1 < nbins) lbase[i1] = run0 + n0;
            if (lane == 63u) s_total = inc;
            if (n0) g0 = w.offset[first + i0] + atomicAdd(&w.cursor[first + i0], n0);
            if (n1) g1 = w.offset[first + i1] + atomicAdd(&w.cursor[first + i1], n1);
        } else {   // many bins per lane: consecutive runs of bins, serial inside the lane
            const uint32_t per = (nbins + 63u) / 64u, lo = lane * per, hi = min(nbins, lo + per);
            uint32_t sum = 0;
            for (uint32_t i = lo; i < hi; i++) sum += hist[i];
            uint32_t inc = sum;
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                const uint32_t up = __shfl_up(inc, d, 64);
                if (lane >= d) inc += up;
            }
            uint32_t run = inc - sum;
            for (uint32_t i = lo; i < hi; i++) {
                const uint32_t n = hist[i];
                lbase[i] = run;
                delta[i] = n ? w.offset[first + i] + atomicAdd(&w.cursor[first + i], n) - run : 0u;
                run += n;
            }
            if (lane == 63u) s_total = inc;
        }
    }
#ifndef NGP_STAMP_FILL
    __syncthreads();
#else
    {   // (the stamp's vmcnt(0) would wait for the cursor atomics here: stamp without it)
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long st_now;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory");
        if (threadIdx.x == 0) atomicAdd(&ngp_dbg_fill_stamps[level][5], st_now - st_prev);   // scan + cursor atomics issued
        st_prev = st_now;
        __syncthreads();
    }
#endif
    if (threadIdx.x < 64u && fast) {
        const uint32_t i0 = 2u * lane, i1 = i0 + 1u;
        if (i0 < nbins) delta[i0] = g0 - run0;
        if (i1 < nbins) delta[i1] = g1 - (run0 + n0);
    }

    // stage the records sorted by chunk
    float gmax = nan ? __uint_as_float(0x7f800000u) : 0.0f;   // NaN -> inf
    if (emit) {
#pragma unroll
        for (uint32_t corner = 0; corner < 8; corner++) {
            gmax = fmaxf(gmax, fmaxf(fabsf(vx[corner]), fabsf(vy[corner])));
            const uint32_t bin = rows[corner] >> kChunkShift;
            const uint32_t slot = lbase[bin] + pos[corner];
            stage_key[slot] = (rows[corner] & (kChunkRows - 1u)) | (bin << kChunkShift);
            stage_val[slot] = make_float2(vx[corner], vy[corner]);
        }
    }
    // largest |gradient| of the call -> fixed-point scale of the reduce kernel; one word for the whole grid,
    // only waves that would raise it touch it
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, d, 64));
    if ((threadIdx.x & 63u) == 0 && __float_as_uint(gmax) > __builtin_nontemporal_load(&w.chunk_base[kMaxLevels + 1]))
        atomicMax(&w.chunk_base[kMaxLevels + 1], __float_as_uint(gmax));
    __syncthreads();
    NGP_FILL_STAMP(6);   // cursor round trip + staging, until the last wave is there
    // consecutive lanes -> consecutive records of one chunk (until the chunk changes): contiguous 8-byte and 2-byte stores
    const uint32_t total = s_total;
    for (uint32_t j = threadIdx.x; j < total; j += kFillBlock) {
        const uint32_t key = stage_key[j];
        const uint32_t dst = j + delta[key >> kChunkShift];
        w.vals[dst] = stage_val[j];
        w.keys[dst] = (uint16_t)(key & (kChunkRows - 1u));
    }
    NGP_FILL_STAMP(7);   // stream-out (stores retired)
}

// ------------------------------------------------------------------ fill, tile-local layout
// The same traversal, staging and record format -- but the workgroup writes into a region of its own (slot j of the
// staging area goes to slot j of the region: the stream-out is a plain copy) and publishes one directory word per chunk.
// No cursor atomics (their issue and round trip were half of a workgroup's life: tools/fill_stamps.py), no counting pass in
// the encoder's forward, no scan; zero-gradient runs still emit nothing.
// LDS: hist[nbins] | lbase[nbins] | stage_val[kRegion] float2 | stage_key[kRegion] u16
__global__ __launch_bounds__(kFillBlock) void bin_fill_local_kernel(
    const float *__restrict__ grad, const float *__restrict__ inputs, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ B_dev, uint32_t B_cap, uint32_t gstride, uint32_t ntiles, uint32_t nbins_cap, LevelRes lv,
    uint32_t gridtype, bool align_corners, uint32_t interp, WsLayout w, uint32_t *__restrict__ dir, uint32_t n_tail,
    MlpDwReduce tail, uint32_t snake_levels, const int32_t *__restrict__ sample_index, uint32_t persistent = 0)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (blockIdx.x < n_tail) {   // passengers: the tiny MLPs' weight-gradient reduction (see bin_fill_kernel)
        float(*part)[64] = reinterpret_cast<float(*)[64]>(lds) + 4 * (threadIdx.x >> 8);
        mlp_reduce_dw_group(tail, blockIdx.x * 2u + (threadIdx.x >> 8), threadIdx.x & 255u, part);
        return;
    }
    uint32_t *hist = lds, *lbase = lds + nbins_cap;
    float2 *stage_val = reinterpret_cast<float2 *>(lds + 2 * nbins_cap);            // 8-byte aligned: nbins_cap % 4 == 0
    uint16_t *stage_key = reinterpret_cast<uint16_t *>(stage_val + kRegion);
    __shared__ uint32_t s_total;
    __shared__ float s_wmax[kFillBlock / 64];

    const uint32_t B = B_dev ? min((uint32_t)max(B_dev[0], 0), B_cap) : B_cap;
    // one work item = 512 samples of one level (the body keeps the indentation it had as the kernel's own)
    auto do_item = [&](uint32_t level, uint32_t tile) {
    const uint32_t b0 = tile * kFillTile;
    if (b0 >= B) return;
    const uint32_t b = b0 + threadIdx.x;
    // position and gradient do not depend on each other: both requests leave before anything waits -- before the level's
    // geometry is read, too (scalar loads the vector requests need not queue behind)
    // (sample_index: the call runs over a LIST of B samples -- the position of entry b is that of sample sample_index[b],
    // the gradient slab is in list order; ngp_x_mlp_backward_list wrote it that way)
    float x[3] = {0.f, 0.f, 0.f};
    float2 gr = make_float2(0.0f, 0.0f);
    if (b < B) {
        const size_t sb = sample_index ? (size_t)(uint32_t)sample_index[b] : (size_t)b;
        gr = reinterpret_cast<const float2 *>(grad)[(size_t)level * gstride + b];
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) x[d] = inputs[sb * 3 + d];
    }
    const Geom<3> g = make_geom<3>(offsets, level, lv.res[level], gridtype);
    // the level's chunks: the table's geometry, no workspace header involved
    uint32_t first = 0;
    for (uint32_t l = 0; l < level; l++) first += ((uint32_t)(offsets[l + 1] - offsets[l]) + kChunkRows - 1) >> kChunkShift;
    const uint32_t nbins = (g.T + kChunkRows - 1) >> kChunkShift;
    NGP_FILL_STAMP_DECL;
    NGP_FILL_STAMP(0);
    for (uint32_t i = threadIdx.x; i < nbins; i += kFillBlock) hist[i] = 0;
    __syncthreads();
    NGP_FILL_STAMP(1);   // (diagnostic build: the stamp waits for the loads -> phase 1 = header + loads)

    // Two compilations of the rest of the kernel, picked per workgroup (the level is uniform): FAST = a hashed level whose
    // table is a power of two with res <= 4096 -- the geometry's kind is then a compile-time fact (no per-corner branches on
    // hashed / dense / modulo), and the x-neighbours of a corner pair always share a chunk (the PAIRS below); the generic
    // one serves the dense levels and every other table.
    const bool fast = g.hashed && g.mode == 1u && g.res <= kChunkRows && kPrimes[0] == 1u;
    auto body = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        Geom<3> gk = g;
        if constexpr (FAST) {
            gk.hashed = true;
            gk.mode = 1u;
        }
        uint32_t rows[8], pos[8];
        float vx[8], vy[8];
        Cell<3> cl = {};
        const bool live = b < B && locate<3>(x, gk.res, align_corners, interp, cl);
        if (!live) gr = make_float2(0.0f, 0.0f);
        const bool nan = !(gr.x == gr.x && gr.y == gr.y);
        const bool active = live && (gr.x != 0.0f || gr.y != 0.0f);
    #pragma unroll
        for (uint32_t corner = 0; corner < 8; corner++) vx[corner] = vy[corner] = 0.0f;
        if (active) {
    #pragma unroll
            for (uint32_t corner = 0; corner < 8; corner++) {
                float wgt = 1.0f;
    #pragma unroll
                for (uint32_t d = 0; d < 3; d++) wgt *= (corner & (1u << d)) ? cl.f[d] : 1.0f - cl.f[d];
                vx[corner] = wgt * gr.x;
                vy[corner] = wgt * gr.y;
            }
        }
        bool emit = active;
        if (mergeable(gk, w)) {   // run merging, as in bin_fill_kernel
            uint32_t dist;
            const bool tail_lane = run_shape(live ? cell_key(cl) : kDeadKey, live, dist);
            const RunFlags flags = run_flags(dist);
            const float active_in_run = run_sum(active ? 1.0f : 0.0f, flags);
            emit = tail_lane && active_in_run > 0.0f;
            if (__ballot(active) != 0ull) {
#pragma unroll
                for (uint32_t corner = 0; corner < 8; corner++) {
                    vx[corner] = run_sum(vx[corner], flags);
                    vy[corner] = run_sum(vy[corner], flags);
                }
            }
        }
        // PAIRS.  On a hashed level whose table is a power of two the x term of the hash is the cell's x coordinate itself
        // (kPrimes[0] = 1), below 2^12 for res <= 4096: the two x-neighbours of a corner pair differ in the low 12 bits only, i.e.
        // they ALWAYS fall into the same 4096-row chunk.  So the eight records of a sample are four pairs: one histogram atomic,
        // one offset read, one 16-byte payload store and one 4-byte key store per PAIR -- half the LDS instructions of the
        // record-by-record path, which stays for the dense levels (their x-pairs straddle a chunk boundary now and then).
        constexpr bool pairs = FAST;
        const uint32_t unit = pairs ? 2u : 1u;   // records per histogram count
        if (emit) {
            const AxisTerms<3> terms = axis_terms<3>(gk, cl);
    #pragma unroll
            for (uint32_t corner = 0; corner < 8; corner++) rows[corner] = row_from_terms<3>(gk, terms, corner);
            if (pairs) {
    #pragma unroll
                for (uint32_t q = 0; q < 4; q++) pos[q] = atomicAdd(&hist[rows[2 * q] >> kChunkShift], 1u);
            } else {
    #pragma unroll
                for (uint32_t corner = 0; corner < 8; corner++) pos[corner] = atomicAdd(&hist[rows[corner] >> kChunkShift], 1u);
            }
        }
        NGP_FILL_STAMP(3);
        __syncthreads();
        NGP_FILL_STAMP(4);

        // exclusive scan of the histogram (each chunk's count rounded up to a quad of records) -> staging offsets = offsets
        // inside the tile's region.  Wave 0 scans the level's <= kLocalBins bins, kLocalBins / 64 per lane, and leaves lbase[] (in
        // histogram units); directory words and the null records that pad a run go out AFTER the barrier, one bin per lane of the
        // whole workgroup -- the seven other waves used to wait while wave 0 issued up to 128 scattered 4-byte stores.
        const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
        const uint32_t quad = 4u / unit;          // histogram counts per quad of records
        if (wid == 0) {
            constexpr uint32_t K = kLocalBins / 64u;   // consecutive bins per lane
            uint32_t n[K], q[K], sum = 0;
    #pragma unroll
            for (uint32_t j = 0; j < K; j++) {
                const uint32_t i = K * lane + j;
                n[j] = i < nbins ? hist[i] : 0u;
                q[j] = (n[j] + quad - 1u) & ~(quad - 1u);
                sum += q[j];
            }
            uint32_t inc = sum;
    #pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                const uint32_t up = __shfl_up(inc, d, 64);
                if (lane >= d) inc += up;
            }
            uint32_t run = inc - sum;
    #pragma unroll
            for (uint32_t j = 0; j < K; j++) {
                const uint32_t i = K * lane + j;
                if (i < nbins) lbase[i] = run;
                run += q[j];
            }
            if (lane == 63u) s_total = inc * unit;
        }
        NGP_FILL_STAMP(5);   // scan
        __syncthreads();

        uint32_t *dcol = dir + (size_t)first * ntiles + tile;   // dir[chunk][tile]
        if (threadIdx.x < nbins) {   // (nbins <= kLocalBins <= kFillBlock)
            const uint32_t i = threadIdx.x, n = hist[i], base = lbase[i];
            dcol[(size_t)i * ntiles] = (base * unit) | ((n * unit) << 16);
            if (n & (quad - 1u)) {   // pad the run to a quad with null records (key 0, value 0)
                for (uint32_t k = n * unit; k < ((n + quad - 1u) & ~(quad - 1u)) * unit; k++) {
                    stage_key[base * unit + k] = 0;
                    stage_val[base * unit + k] = make_float2(0.f, 0.f);
                }
            }
        }
        float gmax = nan ? __uint_as_float(0x7f800000u) : 0.0f;   // NaN -> inf
        if (emit) {
    #pragma unroll
            for (uint32_t corner = 0; corner < 8; corner++) gmax = fmaxf(gmax, fmaxf(fabsf(vx[corner]), fabsf(vy[corner])));
            if (pairs) {
                float4 *pv = reinterpret_cast<float4 *>(stage_val);
                uint32_t *pk = reinterpret_cast<uint32_t *>(stage_key);
    #pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const uint32_t slot = lbase[rows[2 * q] >> kChunkShift] + pos[q];
                    pk[slot] = (rows[2 * q] & (kChunkRows - 1u)) | ((rows[2 * q + 1] & (kChunkRows - 1u)) << 16);
                    pv[slot] = make_float4(vx[2 * q], vy[2 * q], vx[2 * q + 1], vy[2 * q + 1]);
                }
            } else {
    #pragma unroll
                for (uint32_t corner = 0; corner < 8; corner++) {
                    const uint32_t slot = lbase[rows[corner] >> kChunkShift] + pos[corner];
                    stage_key[slot] = (uint16_t)(rows[corner] & (kChunkRows - 1u));
                    stage_val[slot] = make_float2(vx[corner], vy[corner]);
                }
            }
        }
    #pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, d, 64));
        if (lane == 0) s_wmax[wid] = gmax;
        __syncthreads();
        // largest |gradient| of the call -> fixed-point scale of the reduce kernel: one word for the whole grid, touched by one
        // lane per workgroup and only when the workgroup would raise it.  The current value is read with a device-scope atomic
        // load: a plain (or "non-temporal") load of this uniform address becomes a scalar load, and the scalar cache never
        // sees the other workgroups' atomics -- every wave then believes the word is still zero and the kernel degenerates
        // into 50 000 serialised same-address atomics (600 us instead of 100).
        if (threadIdx.x == 0) {
            float m = s_wmax[0];
    #pragma unroll
            for (uint32_t k = 1; k < kFillBlock / 64; k++) m = fmaxf(m, s_wmax[k]);
            uint32_t *word = &w.chunk_base[kMaxLevels + 1];
            if (__float_as_uint(m) > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(word, __float_as_uint(m));
        }
        NGP_FILL_STAMP(6);
        // the region is the staging area, slot for slot: 16-byte stores of two payloads, 8-byte stores of four keys
        const uint32_t total = s_total;
        const size_t region = ((size_t)level * ntiles + tile) * kRegion;   // (kRegion: a multiple of 4 records)
        float4 *dv = reinterpret_cast<float4 *>(w.vals + region);
        const float4 *sv = reinterpret_cast<const float4 *>(stage_val);
        for (uint32_t j = threadIdx.x; j < total / 2u; j += kFillBlock) dv[j] = sv[j];      // (total is a multiple of 4)
        uint2 *dk = reinterpret_cast<uint2 *>(w.keys + region);
        const uint2 *sk = reinterpret_cast<const uint2 *>(stage_key);
        for (uint32_t j = threadIdx.x; j < total / 4u; j += kFillBlock) dk[j] = sk[j];
        NGP_FILL_STAMP(7);
    };
    if (fast)
        body(std::true_type{});
    else
        body(std::false_type{});
    };
    if (persistent) {
        // persistent form (as the slab forward's, engine_kernels.hip): the list is as long as the samples the device counts,
        // not as the capacity the launch was sized for; the workgroups of one XCD stride through that XCD's levels
        const uint32_t bid = blockIdx.x - n_tail, k = bid & 7u, G8 = (gridDim.x - n_tail) >> 3;
        const uint32_t tiles = (B + kFillTile - 1u) / kFillTile, len = snake_rounds(snake_levels) * tiles;
        for (uint32_t it = bid >> 3; it < len; it += G8) {
            const uint32_t round = it / tiles, t = it - round * tiles;
            const uint32_t lvl = (round >> 1) * 16u + ((round & 1u) ? 15u - k : k);
            if (lvl < snake_levels) do_item(lvl, t);
            __syncthreads();   // (the next item reuses the histogram and the staging area)
        }
        return;
    }
    uint32_t level, tile;
    if (snake_levels) {
        snake_level_tile(blockIdx.x - n_tail, ntiles, snake_levels, level, tile);
        if (level == kNoLevel) return;
    } else {
        const uint32_t item = xcd_remap(blockIdx.x - n_tail, gridDim.x - n_tail);
        level = item / ntiles;
        tile = item - level * ntiles;
    }
    do_item(level, tile);
}

// ------------------------------------------------------------------ reduce
// Optimiser state for the fused variant: the chunk's gradient never leaves LDS, Adam (torch.optim.Adam, as in
// engine_kernels.hip: adam_span) is applied to the chunk's rows right there.  Needs one segment per chunk.
struct AdamArgs {
    float *param, *exp_avg, *exp_avg_sq;
    const float *hyper;   // {lr, 1 - beta1^t, 1/sqrt(1 - beta2^t)}
    float b1, b2, eps;
};

// MODE 0: grad[chunk] += sums   1: Adam on the chunk's rows (no gradient written)   2: grad[chunk] = sums, zeros included
// MODE 3: as 2, but the gradient is stored as bfloat16 (round to nearest even): the data-parallel wire format
// (modes 1 and 2 need one segment per chunk and visit every chunk, also the ones without records)
// LOCAL: the records live in per-tile regions (bin_fill_local_kernel); the chunk's runs are found through its directory
// column.  One workgroup per chunk in every mode (the += of mode 0 needs no atomics then).
// waves per SIMD the reduce kernel is compiled for: 8 per workgroup, as many workgroups per CU as its LDS allows
constexpr uint32_t kReduceWaves = kChunkShift <= 11 ? 6 : 4;
#ifndef NGP_REDUCE_PREFETCH
#define NGP_REDUCE_PREFETCH 0
#endif
struct LocalRecords {
    const uint32_t *dir = nullptr;    // [n_chunks][ntiles]: first slot | count << 16
    const int32_t *B_dev = nullptr;
    uint32_t B_cap = 0, ntiles = 0;
    uint32_t chunk_lo = 0, chunk_hi = 0xffffffffu;   // the launch's chunks (level-major numbering); default: all
};
// Dynamic loss scale (mlp_common.hpp: LossScalerWord).  The reduce launch is the first one that knows whether the step saw a
// non-finite gradient anywhere: the weight-gradient reduction (passengers of the fill launch in front of it) has raised
// LS_FOUND for the MLPs, the fill has left the batch's largest feature gradient in the workspace header.  The fused-Adam
// reduce skips the table on either; the first n_pass workgroups are passengers that do the MLP weights' Adam step (and their
// operand-image entries) under the same verdict, and one of them records the table's half of it for step_begin.
struct ScalerTail {
    float *scaler = nullptr;
    uint32_t n_pass = 0;
    MlpDwReduce mlp{};
};
constexpr uint32_t kAdamPassengers = 2 * kAccFloats / kReduceBlock;

template <int MODE, bool LOCAL = false>
__global__ __launch_bounds__(kReduceBlock) __attribute__((amdgpu_waves_per_eu(kReduceWaves, kReduceWaves))) void bin_reduce_kernel(const int32_t *__restrict__ offsets,
                                                                 float *__restrict__ grad_table, uint32_t L, WsLayout w,
                                                                 AdamArgs opt, LocalRecords loc = LocalRecords{},
                                                                 ScalerTail st = ScalerTail{})
{
    constexpr bool ADAM = MODE == 1, ALL = MODE != 0 || LOCAL;
    uint32_t *sw = reinterpret_cast<uint32_t *>(st.scaler);
    if (sw && (blockIdx.x < st.n_pass || (st.n_pass == 0 && blockIdx.x == 0 && threadIdx.x == 0))) {
        // (with passengers: workgroup 0 is one; without -- the exchange step, whose optimiser kernels read the word --
        // thread 0 of the first chunk's workgroup writes it and carries on)
        const bool overflow = !(__uint_as_float(w.chunk_base[kMaxLevels + 1]) < __uint_as_float(0x7f800000u));
        if (blockIdx.x == 0 && threadIdx.x == 0 && overflow) sw[LS_FOUND] = 1u;
        if (st.n_pass != 0) {
            if (!overflow && sw[LS_FOUND] == 0u && st.mlp.adam.param)
                mlp_adam_group(st.mlp, blockIdx.x * kReduceBlock + threadIdx.x);
            return;
        }
    }
    // (the chunk workgroups test the verdict further down, next to the batch maximum they read anyway: its round trip then
    // hides behind the directory and optimiser-state requests instead of standing in front of them)
    __shared__ unsigned long long acc[kChunkRows * 2];   // 16 bytes a row: int64 fixed-point sums, [row][channel]
    __shared__ uint32_t s_chunk, s_level;
    __shared__ uint32_t s_base[kMaxLevels + 1];
    __shared__ uint32_t s_pref[LOCAL ? kMaxTiles + 1 : 1], s_wsum[kReduceBlock / 64];
    __shared__ uint16_t s_start[LOCAL ? kMaxTiles : 1];
    // header words the whole workgroup needs: fetched once, side by side (chains of dependent global loads -- a binary
    // search, a level walk -- cost more than the chunk's arithmetic)
    if (LOCAL) {   // the chunks of each level follow from the table's geometry
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (uint32_t l = 0; l < L; l++) {
                s_base[l] = run;
                run += ((uint32_t)(offsets[l + 1] - offsets[l]) + kChunkRows - 1) >> kChunkShift;
            }
            s_base[L] = run;
        }
    } else if (threadIdx.x <= L) {
        s_base[threadIdx.x] = w.chunk_base[threadIdx.x];
    }
    for (uint32_t i = threadIdx.x; i < kChunkRows * 2; i += kReduceBlock) acc[i] = 0ull;
    __syncthreads();
    const uint32_t n_chunks = s_base[L];
    const uint32_t item = blockIdx.x - st.n_pass + loc.chunk_lo;
    if (ALL) {   // one segment per chunk: the work item IS the chunk
        if (item >= min(n_chunks, loc.chunk_hi)) return;
    } else {
        if (item >= w.seg_base[n_chunks]) return;
    }
    if (threadIdx.x == 0) {
        uint32_t lo = item;
        if (!ALL) {   // largest chunk with seg_base[chunk] <= item
            lo = 0;
            uint32_t hi = n_chunks;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (w.seg_base[mid] <= item)
                    lo = mid;
                else
                    hi = mid;
            }
        }
        uint32_t level = 0;
        while (level + 1 < L && s_base[level + 1] <= lo) level++;
        s_chunk = lo;
        s_level = level;
    }
    __syncthreads();
    const uint32_t chunk = s_chunk;
    uint32_t n_seg = 1, beg = 0, end = 0;
    if constexpr (!LOCAL) {
        const uint32_t reserved = w.count[chunk];   // records counted for the chunk (what the segments were cut from)
        const uint32_t cnt = w.cursor[chunk];       // records the fill kernel really wrote (zero-gradient runs emit none)
        n_seg = w.seg_base[chunk + 1] - w.seg_base[chunk];
        const uint32_t seg_len = n_seg == 1 ? max(reserved, 1u) : (reserved > kSegBig ? kSegBig : kSeg);
        const uint32_t seg = item - w.seg_base[chunk];
        beg = w.offset[chunk] + seg * seg_len;                  // multiple of 4 records
        end = min(w.offset[chunk] + cnt, beg + seg_len);
        if (!ALL && beg >= end) return;   // empty chunk: nothing to add (modes 1 and 2 still have rows to write)
    }

    // which level does this chunk belong to, and where does it start in the table
    const uint32_t level = s_level;
    const uint32_t row0 = (chunk - s_base[level]) << kChunkShift;
    const uint32_t T = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t rows_here = min(kChunkRows, T - row0);
    // fused variant: the optimiser state of this lane's rows is requested now, so that its latency hides behind the
    // record streaming (it does not depend on the gradient)
    // tile-local records: the chunk's directory column is requested first -- in front of the optimiser state, whose 96 KiB
    // per workgroup would otherwise have to arrive before the (in-order) directory words count as returned
    constexpr uint32_t kDirPerLane = LOCAL ? kMaxTiles / kReduceBlock : 1;
    uint32_t dirw[kDirPerLane];
    uint32_t tiles = 0;
    if constexpr (LOCAL) {
        const uint32_t Bn = loc.B_dev ? min((uint32_t)max(loc.B_dev[0], 0), loc.B_cap) : loc.B_cap;
        tiles = (Bn + kFillTile - 1) / kFillTile;
        const uint32_t *dcol = loc.dir + (size_t)chunk * loc.ntiles;
#pragma unroll
        for (uint32_t q = 0; q < kDirPerLane; q++) {
            const uint32_t t = threadIdx.x * kDirPerLane + q;
            dirw[q] = t < tiles ? dcol[t] : 0u;
        }
    }
    constexpr uint32_t kPer = kChunkRows / kReduceBlock;
    float2 pp[kPer], mm[kPer], vv[kPer];
    const size_t adam_base = (size_t)(uint32_t)offsets[level] + row0;
    // When is the optimiser state requested?  NGP_REDUCE_PREFETCH = 1: in front of the record gather (its latency hides
    // behind the gather, but its 48 registers a lane next to the gather's made the compiler spill 12 at the 128 a lane that
    // two workgroups per CU allow); 0 (default): behind the gather -- 102 registers, no spill, the other workgroup of the CU
    // covers the latency: table backward 142 → 135 us in the step, same-box pairs; 2: the parameter in front, the moments behind.
    constexpr int kPrefetch = NGP_REDUCE_PREFETCH;
    auto load_state = [&](bool param, bool moments) {
        const float2 *p2 = reinterpret_cast<const float2 *>(opt.param) + adam_base,
                     *m2 = reinterpret_cast<const float2 *>(opt.exp_avg) + adam_base,
                     *v2 = reinterpret_cast<const float2 *>(opt.exp_avg_sq) + adam_base;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint32_t i = threadIdx.x + j * kReduceBlock;
            if (i < rows_here) {
                if (param) pp[j] = p2[i];
                if (moments) {
                    // (the moments are touched once per step, here: streamed past the caches, so that the 98 MB of them do
                    // not evict the table the next forward pass gathers from -- measured -0.75 % step time)
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const f2 tm = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(m2 + i));
                    const f2 tv = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(v2 + i));
                    mm[j] = make_float2(tm.x, tm.y);
                    vv[j] = make_float2(tv.x, tv.y);
                }
            }
        }
    };
    if (ADAM && kPrefetch != 0) load_state(true, kPrefetch == 1);

    // fixed-point scale: max |g| < 2^e  ->  |g * 2^k| < 2^(62 - headroom) with k = 62 - headroom - e
    int e;
    const float gmax = __uint_as_float(w.chunk_base[kMaxLevels + 1]);
    // a non-finite gradient anywhere in the batch (the fill turns NaN into inf): no fixed-point scale exists and an Adam
    // step on it would poison the table for good -- skip the table's update, as the reference's GradScaler skips the step
    // (the MLP weights are protected element by element: mlp_reduce_dw_group)
    // (dynamic loss scale: the overflow word the weight-gradient reduction may have raised counts the same; workgroup 0 only
    // ever writes it when the maximum is non-finite, which every workgroup sees for itself)
    if (ADAM && (!(gmax < __uint_as_float(0x7f800000u)) || (sw && sw[LS_FOUND] != 0u))) return;
    frexpf(gmax, &e);
    const int k = 62 - kHeadroomBits - e;

    // 4 records per lane and load group: keys 8 bytes, payloads 2 x 16 bytes; two groups in flight
    auto apply = [&](uint32_t row, float gx, float gy) {
        const long long qx = __double2ll_rn(scalbn((double)gx, k));
        const long long qy = __double2ll_rn(scalbn((double)gy, k));
        atomicAdd(&acc[row * 2], (unsigned long long)qx);
        atomicAdd(&acc[row * 2 + 1], (unsigned long long)qy);
    };
    if constexpr (LOCAL) {
        // The chunk's records: one run per tile, found through the directory column dir[chunk][0 .. tiles) -- fetched above
        // (one round trip), kept in LDS as a prefix of run lengths (in quads) + the runs' first slots; after that every load
        // is independent of every other.  Runs start at multiples of four slots and are padded with null records: a quad
        // is one 8-byte load of keys and two 16-byte loads of payloads, like the streams of the global-bins layout.
        // (more than kMaxTiles live tiles -- over a million samples -- are taken in batches of kMaxTiles)
        const uint32_t all_tiles = tiles;
        for (uint32_t tb = 0; tb == 0 || tb < all_tiles; tb += kMaxTiles) {
        tiles = min(kMaxTiles, all_tiles - min(tb, all_tiles));
        if (tb != 0) {
            __syncthreads();   // everybody is done with the previous batch's prefix
            const uint32_t *dcol = loc.dir + (size_t)chunk * loc.ntiles + tb;
#pragma unroll
            for (uint32_t q = 0; q < kDirPerLane; q++) {
                const uint32_t t = threadIdx.x * kDirPerLane + q;
                dirw[q] = t < tiles ? dcol[t] : 0u;
            }
        }
        const size_t level_region = ((size_t)level * loc.ntiles + tb) * kRegion;
        constexpr uint32_t kPerLane = kDirPerLane;   // consecutive tiles per lane in the scan
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < kPerLane; q++) sum += ((dirw[q] >> 16) + 3u) >> 2;
        const uint32_t ln = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        uint32_t inc = sum;
#pragma unroll
        for (uint32_t dd = 1; dd < 64u; dd <<= 1) {
            const uint32_t up = __shfl_up(inc, dd, 64);
            if (ln >= dd) inc += up;
        }
        if (ln == 63u) s_wsum[wv] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (uint32_t k = 0; k < wv; k++) run += s_wsum[k];
#pragma unroll
        for (uint32_t q = 0; q < kPerLane; q++) {
            const uint32_t t = threadIdx.x * kPerLane + q;
            s_pref[t] = run;
            s_start[t] = (uint16_t)(dirw[q] & 0xffffu);
            run += ((dirw[q] >> 16) + 3u) >> 2;
        }
        if (threadIdx.x == kReduceBlock - 1) s_pref[kMaxTiles] = run;
        __syncthreads();
        const uint32_t Q = s_pref[kMaxTiles];   // quads of the chunk
        const uint2 *kq = reinterpret_cast<const uint2 *>(w.keys);       // 4 keys per uint2
        const float4 *vq = reinterpret_cast<const float4 *>(w.vals);     // 2 payloads per float4
        auto apply_quad = [&](const uint2 &k4, const float4 &a0, const float4 &a1) {   // (null records add zero to row 0)
            apply(k4.x & 0xffffu, a0.x, a0.y);
            apply(k4.x >> 16, a0.z, a0.w);
            apply(k4.y & 0xffffu, a1.x, a1.y);
            apply(k4.y >> 16, a1.z, a1.w);
        };
        if (Q > 12u * tiles) {
            // a heavy chunk (the few chunks of the coarse levels collect hundreds of records from every tile): a wave streams
            // a run -- 256 records per load round -- two (four) runs per turn.  (Sharing such a chunk among several workgroups was
            // tried -- parked 64-bit sums, a ticket, the last one finishes -- and is slower: the device-scope release /
            // acquire fences write back and invalidate a whole L2 in the middle of a kernel that streams the optimiser state.)
            constexpr uint32_t kRuns = (ADAM || kReduceWaves > 4) ? 2 : 4;   // (register budget: optimiser state, or 80 a lane)
            for (uint32_t t0 = wv; t0 < tiles; t0 += kRuns * (kReduceBlock / 64u)) {
                uint32_t nq[kRuns];
                size_t rq[kRuns];   // first quad of the run
                uint2 k4[kRuns];
                float4 a0[kRuns], a1[kRuns];
#pragma unroll
                for (uint32_t u = 0; u < kRuns; u++) {
                    const uint32_t t = min(t0 + u * (kReduceBlock / 64u), kMaxTiles - 1u);
                    nq[u] = s_pref[t + 1] - s_pref[t];
                    rq[u] = (level_region + (size_t)t * kRegion + s_start[t]) >> 2;
                    k4[u] = make_uint2(0, 0);
                    a0[u] = a1[u] = make_float4(0, 0, 0, 0);
                    if (ln < nq[u]) {
                        k4[u] = kq[rq[u] + ln];
                        a0[u] = vq[(rq[u] + ln) * 2];
                        a1[u] = vq[(rq[u] + ln) * 2 + 1];
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kRuns; u++) {
                    if (ln < nq[u]) apply_quad(k4[u], a0[u], a1[u]);
                    for (uint32_t j = ln + 64u; j < nq[u]; j += 64u)   // runs longer than 256 records
                        apply_quad(kq[rq[u] + j], vq[(rq[u] + j) * 2], vq[(rq[u] + j) * 2 + 1]);
                }
            }
        } else {
            // light chunks (the hashed levels: 10-30 records per run): a flat index over the quads of the concatenated runs,
            // two independent quads per lane and turn; a binary search over the LDS prefix maps a quad to its tile
            const uint32_t top = 1u << (32 - __clz(max(tiles, 2u) - 1u));   // power of two >= tiles (<= kMaxTiles)
            auto locate_quad = [&](uint32_t v) -> size_t {   // tile = largest t with s_pref[t] <= v
                uint32_t lo = 0;
                for (uint32_t step = top >> 1; step >= 1; step >>= 1)
                    if (s_pref[lo + step] <= v) lo += step;
                return ((level_region + (size_t)lo * kRegion + s_start[lo]) >> 2) + (v - s_pref[lo]);
            };
            for (uint32_t v0 = threadIdx.x; v0 < Q; v0 += 2u * kReduceBlock) {
                const uint32_t v1 = v0 + kReduceBlock;
                const size_t qa = locate_quad(v0), qb = v1 < Q ? locate_quad(v1) : qa;
                const uint2 ka = kq[qa];
                const float4 a0 = vq[qa * 2], a1 = vq[qa * 2 + 1];
                uint2 kb = make_uint2(0, 0);
                float4 b0 = make_float4(0, 0, 0, 0), b1 = b0;
                if (v1 < Q) {
                    kb = kq[qb];
                    b0 = vq[qb * 2];
                    b1 = vq[qb * 2 + 1];
                }
                apply_quad(ka, a0, a1);
                if (v1 < Q) apply_quad(kb, b0, b1);
            }
        }
        }   // batches of tiles
    }
    const uint2 *key4 = reinterpret_cast<const uint2 *>(w.keys);       // 4 keys per uint2
    const float4 *val2 = reinterpret_cast<const float4 *>(w.vals);     // 2 payloads per float4
    for (uint32_t i0 = beg + threadIdx.x * 4; i0 < end; i0 += kReduceBlock * 8) {
        const uint32_t i1 = i0 + kReduceBlock * 4;
        const uint2 ka = key4[i0 >> 2];
        const float4 a0 = val2[i0 >> 1], a1 = val2[(i0 >> 1) + 1];
        uint2 kb = make_uint2(0, 0);
        float4 b0 = make_float4(0, 0, 0, 0), b1 = b0;
        if (i1 < end) {
            kb = key4[i1 >> 2];
            b0 = val2[i1 >> 1];
            b1 = val2[(i1 >> 1) + 1];
        }
        apply(ka.x & 0xffffu, a0.x, a0.y);
        if (i0 + 1 < end) apply(ka.x >> 16, a0.z, a0.w);
        if (i0 + 2 < end) apply(ka.y & 0xffffu, a1.x, a1.y);
        if (i0 + 3 < end) apply(ka.y >> 16, a1.z, a1.w);
        if (i1 < end) apply(kb.x & 0xffffu, b0.x, b0.y);
        if (i1 + 1 < end) apply(kb.x >> 16, b0.z, b0.w);
        if (i1 + 2 < end) apply(kb.y & 0xffffu, b1.x, b1.y);
        if (i1 + 3 < end) apply(kb.y >> 16, b1.z, b1.w);
    }
    if (ADAM && kPrefetch != 1) load_state(kPrefetch == 0, true);
    __syncthreads();
    float *dst = grad_table + ((size_t)(uint32_t)offsets[level] + row0) * 2;
    auto to_float = [&](unsigned long long q) { return (float)scalbn((double)(long long)q, -k); };
    if (ADAM) {
        float2 *p2 = reinterpret_cast<float2 *>(opt.param) + adam_base, *m2 = reinterpret_cast<float2 *>(opt.exp_avg) + adam_base,
               *v2 = reinterpret_cast<float2 *>(opt.exp_avg_sq) + adam_base;
        const float step_size = opt.hyper[0] / opt.hyper[1], rsqrt_bc2 = opt.hyper[2];
        const float b1 = opt.b1, b2 = opt.b2, eps = opt.eps;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint32_t i = threadIdx.x + j * kReduceBlock;
            if (i < rows_here) {
                const float gx = to_float(acc[i * 2]), gy = to_float(acc[i * 2 + 1]);
                mm[j].x = b1 * mm[j].x + (1.0f - b1) * gx;
                vv[j].x = b2 * vv[j].x + (1.0f - b2) * gx * gx;
                pp[j].x -= step_size * (mm[j].x / (sqrtf(vv[j].x) * rsqrt_bc2 + eps));
                mm[j].y = b1 * mm[j].y + (1.0f - b1) * gy;
                vv[j].y = b2 * vv[j].y + (1.0f - b2) * gy * gy;
                pp[j].y -= step_size * (mm[j].y / (sqrtf(vv[j].y) * rsqrt_bc2 + eps));
                p2[i] = pp[j];
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 tm = {mm[j].x, mm[j].y}, tv = {vv[j].x, vv[j].y};
                __builtin_nontemporal_store(tm, reinterpret_cast<f2 *>(m2 + i));
                __builtin_nontemporal_store(tv, reinterpret_cast<f2 *>(v2 + i));
            }
        }
    } else if (MODE == 2) {
        float2 *d2 = reinterpret_cast<float2 *>(dst);
        for (uint32_t i = threadIdx.x; i < rows_here; i += kReduceBlock)
            d2[i] = make_float2(to_float(acc[i * 2]), to_float(acc[i * 2 + 1]));
    } else if (MODE == 3) {
        uint32_t *d16 = reinterpret_cast<uint32_t *>(grad_table) + ((size_t)(uint32_t)offsets[level] + row0);
        auto bf16 = [](float f) {
            const uint32_t u = __float_as_uint(f);
            return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
        };
        for (uint32_t i = threadIdx.x; i < rows_here; i += kReduceBlock)
            d16[i] = bf16(to_float(acc[i * 2])) | (bf16(to_float(acc[i * 2 + 1])) << 16);
    } else if (n_seg == 1) {
        float2 *d2 = reinterpret_cast<float2 *>(dst);
        for (uint32_t i = threadIdx.x; i < rows_here; i += kReduceBlock) {
            float2 v = d2[i];
            v.x += to_float(acc[i * 2]);
            v.y += to_float(acc[i * 2 + 1]);
            d2[i] = v;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < rows_here * 2; i += kReduceBlock)
            if (acc[i] != 0ull) unsafeAtomicAdd(dst + i, to_float(acc[i]));
    }
}

}  // namespace ngp

using namespace ngp;

#ifdef NGP_STAMP_FILL
extern "C" int ngp_dbg_read_fill_stamps(unsigned long long *out, int reset)
{
    static unsigned long long zero[kMaxLevels * 8] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ngp_dbg_fill_stamps), sizeof(zero)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ngp_dbg_fill_stamps), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

extern "C" size_t ngp_x_grid_backward_workspace_bytes(uint32_t B, uint32_t L, uint32_t n_rows_total)
{
    const uint32_t n_chunks_max = n_rows_total / kChunkRows + L + 1;
    return ws_bytes(B, L, n_chunks_max);
}

namespace {
// record layout: tile-local regions + directory (default) or global bins with counts, scan and cursors (NGP_BINNED_LOCAL=0,
// or a level with more than kLocalBins chunks)
bool binned_local(uint32_t B, uint32_t nbins_cap)
{
    static const bool enabled = [] {
        const char *e = getenv("NGP_BINNED_LOCAL");
        return !(e && e[0] == '0');
    }();
    return enabled && nbins_cap <= kLocalBins && (uint64_t)ceil_div(B, kFillTile) * kMaxLevels * kRegion < (1ull << 40);
}

struct BinnedCall {
    LevelRes lv;
    WsLayout w;
    uint32_t n_chunks_max, nbins_cap;
    size_t fill_lds;
};

// shared argument checks of the three entry points; returns NGP_OK or the error code (message already set)
int binned_setup(BinnedCall &c, const char *who, const int32_t *offsets, uint32_t B, uint32_t L, uint32_t max_level, float S,
                 uint32_t H, uint32_t n_rows_total, uint32_t max_level_rows, void *workspace, size_t workspace_bytes)
{
    NGP_REQUIRE(offsets && workspace, "%s: null tensor", who);
    NGP_REQUIRE(fill_levels(c.lv, S, H, L), "%s: L must be in [1, %u]", who, kMaxLevels);
    NGP_REQUIRE(max_level <= L, "%s: max_level > L", who);
    NGP_REQUIRE(((uintptr_t)workspace & 15u) == 0, "%s: workspace must be 16-byte aligned", who);
    c.n_chunks_max = n_rows_total / kChunkRows + L + 1;
    NGP_REQUIRE(workspace_bytes >= ws_bytes(B, L, c.n_chunks_max), "%s: workspace too small", who);
    NGP_REQUIRE((uint64_t)B * L * 8 + 4ull * c.n_chunks_max < (1ull << 32), "%s: B * L too large", who);
    NGP_REQUIRE(c.n_chunks_max <= kMaxChunks, "%s: table too large (> %u chunks)", who, kMaxChunks);
    c.w = ws_layout(workspace, c.n_chunks_max, ws_rec_cap(B, L, c.n_chunks_max));
    // LDS histograms are per level: the caller may tell us the largest level (rows); 0 = unknown
    NGP_REQUIRE(max_level_rows <= n_rows_total, "%s: max_level_rows > n_rows_total", who);
    const uint32_t level_chunks = max_level_rows ? ceil_div(max_level_rows, kChunkRows) : c.n_chunks_max;
    c.nbins_cap = (level_chunks + 3u) & ~3u;
    c.fill_lds = (size_t)c.nbins_cap * 12 + (size_t)kFillTile * 8 * 12;
    return NGP_OK;
}
}  // namespace

extern "C" int ngp_x_grid_backward_binned_geometry(uint32_t *out)
{
    NGP_REQUIRE(out, "grid_backward_binned_geometry: null pointer");
    out[0] = kChunkRows;
    out[1] = kFillTile;
    out[2] = kRegion;
    out[3] = kMaxTiles;
    return NGP_OK;
}

extern "C" int ngp_x_grid_backward_binned_counts(uint32_t B, uint32_t L, uint32_t n_rows_total, uint32_t max_level_rows)
{
    const uint32_t n_chunks_max = n_rows_total / kChunkRows + L + 1;
    const uint32_t level_chunks = max_level_rows ? ceil_div(max_level_rows, kChunkRows) : n_chunks_max;
    return binned_local(B, (level_chunks + 3u) & ~3u) ? 0 : 1;
}

// positions only: plan + count + scan.  in_bound > 0: `inputs` are world positions (the encoder's [0,1] mapping is
// applied here), so this half can run as soon as the samples exist -- e.g. on another stream, before the forward pass
extern "C" int ngp_x_grid_backward_binned_prepare(const float *inputs, float in_bound, const int32_t *offsets,
                                                  const int32_t *B_dev, uint32_t B, uint32_t L, uint32_t max_level,
                                                  float S, uint32_t H, uint32_t gridtype, int align_corners,
                                                  uint32_t interp, uint32_t n_rows_total, uint32_t max_level_rows,
                                                  int single_segment, uint32_t merge_max_res, int stage, void *workspace,
                                                  size_t workspace_bytes, ngp_stream_t stream)
{
    if (B == 0 || max_level == 0) return NGP_OK;
    NGP_REQUIRE(stage >= 0 && stage <= 2, "grid_backward_binned_prepare: stage must be 0 (all), 1 (plan) or 2 (scan)");
    NGP_REQUIRE(inputs || stage != 0, "grid_backward_binned_prepare: null tensor");
    BinnedCall c;
    const int rc = binned_setup(c, "grid_backward_binned_prepare", offsets, B, L, max_level, S, H, n_rows_total,
                                max_level_rows, workspace, workspace_bytes);
    if (rc != NGP_OK) return rc;
    hipStream_t st = as_stream(stream);
    // stage 1 / 2: the counting in between is done by someone who has the rows anyway (ngp_x_grid_encode_forward_slab)
    if (stage != 2) bin_plan_kernel<<<1, 1024, 0, st>>>(offsets, L, c.n_chunks_max, merge_max_res, c.w);
    if (stage == 0) {
        const uint32_t ct = ceil_div(B, kCountTile);
        bin_count_kernel<<<ct * max_level, kBlock, (size_t)c.nbins_cap * 4, st>>>(inputs, offsets, B_dev, B, ct, c.lv,
                                                                                    gridtype, align_corners != 0, interp,
                                                                                    in_bound, c.w);
    }
    if (stage != 1) bin_scan_kernel<<<1, 1024, 0, st>>>(L, c.w, single_segment != 0);
    NGP_CHECK_LAUNCH("grid_backward_binned_prepare");
    return NGP_OK;
}

// fill + reduce on a prepared workspace (same inputs, in [0,1], and the same B_dev value as the prepare call)
static int binned_apply(const char *who, const MlpDwReduce *tail, const int32_t *sample_index, const float *grad,
                        const float *inputs, const int32_t *offsets,
                                                float *grad_embeddings, const int32_t *B_dev, uint32_t B,
                                                uint32_t grad_stride, uint32_t L, uint32_t max_level, float S,
                                                uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                                uint32_t n_rows_total, uint32_t max_level_rows, void *workspace,
                                                size_t workspace_bytes, float *adam_param, float *adam_exp_avg,
                                                float *adam_exp_avg_sq, const float *adam_hyper, float beta1,
                                                float beta2, float eps, int overwrite, ngp_stream_t stream,
                                                float *scaler = nullptr)
{
    (void)who;
    if (B == 0 || max_level == 0) return NGP_OK;
    const bool fused = adam_param != nullptr;
    NGP_REQUIRE(!(fused && overwrite), "grid_backward_binned_apply: overwrite and fused Adam exclude each other");
    NGP_REQUIRE(!overwrite || max_level == L, "grid_backward_binned_apply: overwrite needs max_level == L");
    NGP_REQUIRE(grad && inputs, "grid_backward_binned_apply: null tensor");
    // grad_embeddings NULL without the fused Adam: FILL ONLY (+ the passengers) -- the caller reduces the chunks in ranges of
    // its own afterwards (ngp_x_grid_backward_binned_reduce_range; tile-local layout)
    const bool fill_only = !grad_embeddings && !fused;
    NGP_REQUIRE(!fused || (adam_exp_avg && adam_exp_avg_sq && adam_hyper && max_level == L),
                "grid_backward_binned_apply: fused Adam needs exp_avg, exp_avg_sq, hyper and max_level == L");
    NGP_REQUIRE(grad_stride >= B, "grid_backward_binned_apply: grad_stride smaller than B");
    BinnedCall c;
    const int rc = binned_setup(c, "grid_backward_binned_apply", offsets, B, L, max_level, S, H, n_rows_total,
                                max_level_rows, workspace, workspace_bytes);
    if (rc != NGP_OK) return rc;
    static const bool lds_ok = [] {   // the fill kernel wants more than the default 64 KiB of dynamic LDS
        return hipFuncSetAttribute(reinterpret_cast<const void *>(bin_fill_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kMaxChunks * 12 + kFillTile * 8 * 12) == hipSuccess;
    }();
    NGP_REQUIRE(lds_ok, "grid_backward_binned_apply: cannot raise the dynamic LDS limit");
    hipStream_t st = as_stream(stream);
    const uint32_t ft = ceil_div(B, kFillTile);
    const uint32_t n_tail = tail ? kDwGroups / 2u : 0u;   // two groups of 64 outputs per 512-lane workgroup
    const AdamArgs opt{adam_param, adam_exp_avg, adam_exp_avg_sq, adam_hyper, beta1, beta2, eps};
    // dynamic loss scale: the reduce launch decides the step (ScalerTail); the MLP weights' Adam step rides on it when the
    // caller asked for one (the fill launch's passengers then only reduce the weight gradients and raise the overflow word)
    ScalerTail stl;
    stl.scaler = scaler;
    if (scaler && tail && tail->adam.param) {
        NGP_REQUIRE(fused, "grid_backward_binned_apply_mlp: a loss scaler with the MLP's Adam step needs the table's fused Adam");
        stl.n_pass = kAdamPassengers;
        stl.mlp = *tail;
    }
    const uint32_t np = stl.n_pass;
    if (binned_local(B, c.nbins_cap)) {   // (else: the global-bins layout below)
        const size_t rec_cap = ws_rec_cap_local(B, L);
        const WsLayout wl = ws_layout(workspace, c.n_chunks_max, rec_cap);
        uint32_t *dir = ws_dir(wl, rec_cap);
        const size_t lds = (size_t)c.nbins_cap * 8 + (size_t)kRegion * 10;
        static const bool snake = !(getenv("NGP_SNAKE") && getenv("NGP_SNAKE")[0] == '0') &&
                                  !(getenv("NGP_SNAKE_FILL") && getenv("NGP_SNAKE_FILL")[0] == '0');   // (as the slab forward)
        const bool sn = snake && max_level >= 8;
        // a device-side sample count under the snake: persistent workgroups, as many per CU as the LDS admits (NGP_FILL_WGS
        // overrides; 0: the capacity-sized grid, whose workgroups are 85 % no-ops late in training -- the live list holds ~ 95 k
        // of 655 k slots -- dealt to the XCDs in between the ones that work: table backward 123-126 -> 112-114 us between the
        // bench's events, 0.2836 -> 0.2702 ms/step; 2, 3 and 4 per CU measure the same)
        static const uint32_t fill_wgs = getenv("NGP_FILL_WGS") ? (uint32_t)atoi(getenv("NGP_FILL_WGS")) : 3u;
        const bool pers = fill_wgs != 0 && sn && B_dev != nullptr && n_tail % 8u == 0;
        const uint32_t blocks = pers ? 256u * min(fill_wgs, 8u) : (sn ? snake_blocks(max_level, ft) : ft * max_level);
        bin_fill_local_kernel<<<blocks + n_tail, kFillBlock, lds, st>>>(
            grad, inputs, offsets, B_dev, B, grad_stride, ft, c.nbins_cap, c.lv, gridtype, align_corners != 0, interp, wl, dir,
            n_tail, tail ? *tail : MlpDwReduce{}, sn ? max_level : 0u, sample_index, pers ? 1u : 0u);
        LocalRecords loc;
        loc.dir = dir;
        loc.B_dev = B_dev;
        loc.B_cap = B;
        loc.ntiles = ft;
        // (levels >= max_level have no records: their directory columns are not read -- the reduce walks L levels only when
        // every level was filled, otherwise max_level of them)
        const uint32_t Lr = max_level;
        if (fill_only) {
            NGP_CHECK_LAUNCH("grid_backward_binned_apply");
            return NGP_OK;
        }
        if (fused)
            bin_reduce_kernel<1, true><<<c.n_chunks_max + np, kReduceBlock, 0, st>>>(offsets, grad_embeddings, Lr, wl, opt, loc, stl);
        else if (overwrite == 2)
            bin_reduce_kernel<3, true><<<c.n_chunks_max + np, kReduceBlock, 0, st>>>(offsets, grad_embeddings, Lr, wl, opt, loc, stl);
        else if (overwrite)
            bin_reduce_kernel<2, true><<<c.n_chunks_max + np, kReduceBlock, 0, st>>>(offsets, grad_embeddings, Lr, wl, opt, loc, stl);
        else
            bin_reduce_kernel<0, true><<<c.n_chunks_max + np, kReduceBlock, 0, st>>>(offsets, grad_embeddings, Lr, wl, opt, loc, stl);
        NGP_CHECK_LAUNCH("grid_backward_binned_apply");
        return NGP_OK;
    }
    NGP_REQUIRE(!sample_index, "grid_backward_binned_apply: a sample list needs the tile-local layout (see ..._binned_counts)");
    NGP_REQUIRE(!fill_only, "grid_backward_binned_apply: fill-only needs the tile-local layout (see ..._binned_counts)");
    bin_fill_kernel<<<ft * max_level + n_tail, kFillBlock, c.fill_lds, st>>>(grad, inputs, offsets, B_dev, B, grad_stride, ft,
                                                                            c.nbins_cap, c.lv, gridtype, align_corners != 0,
                                                                            interp, c.w, n_tail, tail ? *tail : MlpDwReduce{});
    const uint32_t n_items_max = c.n_chunks_max + (uint32_t)(((uint64_t)B * max_level * 8) / kSeg) + 1;
    if (fused)   // prepared with single_segment: one workgroup owns each chunk's rows
        bin_reduce_kernel<1><<<c.n_chunks_max + np, kReduceBlock, 0, st>>>(offsets, grad_embeddings, L, c.w, opt, LocalRecords{}, stl);
    else if (overwrite == 2)
        bin_reduce_kernel<3><<<c.n_chunks_max + np, kReduceBlock, 0, st>>>(offsets, grad_embeddings, L, c.w, opt, LocalRecords{}, stl);
    else if (overwrite)
        bin_reduce_kernel<2><<<c.n_chunks_max + np, kReduceBlock, 0, st>>>(offsets, grad_embeddings, L, c.w, opt, LocalRecords{}, stl);
    else
        bin_reduce_kernel<0><<<n_items_max + np, kReduceBlock, 0, st>>>(offsets, grad_embeddings, L, c.w, opt, LocalRecords{}, stl);
    NGP_CHECK_LAUNCH("grid_backward_binned_apply");
    return NGP_OK;
}

extern "C" int ngp_x_grid_backward_binned_apply(const float *grad, const float *inputs, const int32_t *offsets,
                                                float *grad_embeddings, const int32_t *B_dev, uint32_t B,
                                                uint32_t grad_stride, uint32_t L, uint32_t max_level, float S,
                                                uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                                uint32_t n_rows_total, uint32_t max_level_rows, void *workspace,
                                                size_t workspace_bytes, float *adam_param, float *adam_exp_avg,
                                                float *adam_exp_avg_sq, const float *adam_hyper, float beta1,
                                                float beta2, float eps, int overwrite, float *loss_scaler, ngp_stream_t stream)
{
    return binned_apply("grid_backward_binned_apply", nullptr, nullptr, grad, inputs, offsets, grad_embeddings, B_dev, B, grad_stride, L,
                        max_level, S, H, gridtype, align_corners, interp, n_rows_total, max_level_rows, workspace,
                        workspace_bytes, adam_param, adam_exp_avg, adam_exp_avg_sq, adam_hyper, beta1, beta2, eps, overwrite,
                        stream, loss_scaler);
}

// ngp_x_grid_backward_binned_apply over a LIST of samples (see ngp_x_grid_backward_binned_apply_mlp_list)
extern "C" int ngp_x_grid_backward_binned_apply_list(const float *grad, const float *inputs, const int32_t *sample_index,
                                                     const int32_t *offsets, float *grad_embeddings, const int32_t *B_dev,
                                                     uint32_t B, uint32_t grad_stride, uint32_t L, uint32_t max_level, float S,
                                                     uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                                     uint32_t n_rows_total, uint32_t max_level_rows, void *workspace,
                                                     size_t workspace_bytes, float *adam_param, float *adam_exp_avg,
                                                     float *adam_exp_avg_sq, const float *adam_hyper, float beta1,
                                                     float beta2, float eps, int overwrite, float *loss_scaler,
                                                     ngp_stream_t stream)
{
    return binned_apply("grid_backward_binned_apply", nullptr, sample_index, grad, inputs, offsets, grad_embeddings, B_dev, B,
                        grad_stride, L, max_level, S, H, gridtype, align_corners, interp, n_rows_total, max_level_rows, workspace,
                        workspace_bytes, adam_param, adam_exp_avg, adam_exp_avg_sq, adam_hyper, beta1, beta2, eps, overwrite,
                        stream, loss_scaler);
}

// the same launch sequence with ngp_x_mlp_reduce_dw (same arguments, mlp_ prefix) riding along as extra workgroups of the
// fill kernel: one kernel and one dependent-launch gap fewer on the fused step's critical path
extern "C" int ngp_x_grid_backward_binned_apply_mlp(
    const float *grad, const float *inputs, const int32_t *offsets, float *grad_embeddings, const int32_t *B_dev, uint32_t B,
    uint32_t grad_stride, uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners,
    uint32_t interp, uint32_t n_rows_total, uint32_t max_level_rows, void *workspace, size_t workspace_bytes, float *adam_param,
    float *adam_exp_avg, float *adam_exp_avg_sq, const float *adam_hyper, float beta1, float beta2, float eps, int overwrite,
    uint32_t mlp_M, float mlp_loss_scale, float *dw1, float *dw2, float *dw3, float *dw4, float *dw5, float *dw6,
    const void *mlp_workspace, size_t mlp_workspace_bytes, float *mlp_adam_param, const float *mlp_adam_grad,
    float *mlp_adam_exp_avg, float *mlp_adam_exp_avg_sq, uint32_t mlp_adam_n, const float *mlp_adam_hyper, float mlp_beta1,
    float mlp_beta2, float mlp_eps, void *mlp_adam_image, float *loss_scaler, ngp_stream_t stream)
{
    return ngp_x_grid_backward_binned_apply_mlp_list(
        grad, inputs, nullptr, offsets, grad_embeddings, B_dev, B, grad_stride, L, max_level, S, H, gridtype, align_corners, interp,
        n_rows_total, max_level_rows, workspace, workspace_bytes, adam_param, adam_exp_avg, adam_exp_avg_sq, adam_hyper, beta1,
        beta2, eps, overwrite, mlp_M, mlp_loss_scale, dw1, dw2, dw3, dw4, dw5, dw6, mlp_workspace, mlp_workspace_bytes,
        mlp_adam_param, mlp_adam_grad, mlp_adam_exp_avg, mlp_adam_exp_avg_sq, mlp_adam_n, mlp_adam_hyper, mlp_beta1, mlp_beta2,
        mlp_eps, mlp_adam_image, loss_scaler, stream);
}

// ... over a LIST of samples: entry b of the call is sample sample_index[b] (its position is inputs[sample_index[b]]), the
// gradient slab `grad` is in list order, *B_dev entries are used (NULL = samples 0 .. B - 1: the call above)
extern "C" int ngp_x_grid_backward_binned_apply_mlp_list(
    const float *grad, const float *inputs, const int32_t *sample_index, const int32_t *offsets, float *grad_embeddings,
    const int32_t *B_dev, uint32_t B, uint32_t grad_stride, uint32_t L, uint32_t max_level, float S, uint32_t H,
    uint32_t gridtype, int align_corners, uint32_t interp, uint32_t n_rows_total, uint32_t max_level_rows, void *workspace,
    size_t workspace_bytes, float *adam_param, float *adam_exp_avg, float *adam_exp_avg_sq, const float *adam_hyper,
    float beta1, float beta2, float eps, int overwrite, uint32_t mlp_M, float mlp_loss_scale, float *dw1, float *dw2,
    float *dw3, float *dw4, float *dw5, float *dw6, const void *mlp_workspace, size_t mlp_workspace_bytes,
    float *mlp_adam_param, const float *mlp_adam_grad, float *mlp_adam_exp_avg, float *mlp_adam_exp_avg_sq,
    uint32_t mlp_adam_n, const float *mlp_adam_hyper, float mlp_beta1, float mlp_beta2, float mlp_eps, void *mlp_adam_image,
    float *loss_scaler, ngp_stream_t stream)
{
    NGP_REQUIRE(B != 0 && max_level != 0, "grid_backward_binned_apply_mlp: nothing to launch the reduction with");
    MlpDwReduce r;
    const int rc = mlp_dw_reduce_args(r, "grid_backward_binned_apply_mlp", mlp_M, mlp_loss_scale, dw1, dw2, dw3, dw4, dw5, dw6,
                                      mlp_workspace, mlp_workspace_bytes, mlp_adam_param, mlp_adam_grad, mlp_adam_exp_avg,
                                      mlp_adam_exp_avg_sq, mlp_adam_n, mlp_adam_hyper, mlp_beta1, mlp_beta2, mlp_eps,
                                      mlp_adam_image, loss_scaler);
    if (rc != NGP_OK) return rc;
    return binned_apply("grid_backward_binned_apply_mlp", &r, sample_index, grad, inputs, offsets, grad_embeddings, B_dev, B,
                        grad_stride, L, max_level, S, H, gridtype, align_corners, interp, n_rows_total, max_level_rows, workspace,
                        workspace_bytes, adam_param, adam_exp_avg, adam_exp_avg_sq, adam_hyper, beta1, beta2, eps, overwrite,
                        stream, loss_scaler);
}

// The reduce half on its own, over the chunks [chunk_lo, chunk_hi) of a workspace a fill-only apply call has filled (same B,
// B_dev, L, geometry): grad_embeddings[rows of those chunks] = sums (overwrite 1; 2: bfloat16; 0: +=).  Chunks are numbered
// level-major, ceil(rows of the level / chunk rows) per level (ngp_x_grid_backward_binned_geometry).  The data-parallel step
// reduces the table's levels in two ranges and lets the gradient exchange of the first overlap the reduction of the second.
extern "C" int ngp_x_grid_backward_binned_reduce_range(const int32_t *offsets, float *grad_embeddings, const int32_t *B_dev,
                                                       uint32_t B, uint32_t L, float S, uint32_t H, uint32_t n_rows_total,
                                                       uint32_t max_level_rows, void *workspace, size_t workspace_bytes,
                                                       int overwrite, uint32_t chunk_lo, uint32_t chunk_hi,
                                                       float *loss_scaler, ngp_stream_t stream)
{
    if (B == 0 || chunk_hi <= chunk_lo) return NGP_OK;
    NGP_REQUIRE(offsets && grad_embeddings && workspace, "grid_backward_binned_reduce_range: null tensor");
    BinnedCall c;
    const int rc = binned_setup(c, "grid_backward_binned_reduce_range", offsets, B, L, L, S, H, n_rows_total, max_level_rows,
                                workspace, workspace_bytes);
    if (rc != NGP_OK) return rc;
    NGP_REQUIRE(binned_local(B, c.nbins_cap), "grid_backward_binned_reduce_range: tile-local layout only (see ..._binned_counts)");
    NGP_REQUIRE(chunk_hi <= c.n_chunks_max, "grid_backward_binned_reduce_range: chunk range beyond the table");
    const size_t rec_cap = ws_rec_cap_local(B, L);
    const WsLayout wl = ws_layout(workspace, c.n_chunks_max, rec_cap);
    LocalRecords loc;
    loc.dir = ws_dir(wl, rec_cap);
    loc.B_dev = B_dev;
    loc.B_cap = B;
    loc.ntiles = ceil_div(B, kFillTile);
    loc.chunk_lo = chunk_lo;
    loc.chunk_hi = chunk_hi;
    ScalerTail stl;
    stl.scaler = loss_scaler;   // (every range's first workgroup raises the overflow word on a non-finite batch maximum)
    const AdamArgs opt{};
    hipStream_t st = as_stream(stream);
    const uint32_t n = chunk_hi - chunk_lo;
    if (overwrite == 2)
        bin_reduce_kernel<3, true><<<n, kReduceBlock, 0, st>>>(offsets, grad_embeddings, L, wl, opt, loc, stl);
    else if (overwrite)
        bin_reduce_kernel<2, true><<<n, kReduceBlock, 0, st>>>(offsets, grad_embeddings, L, wl, opt, loc, stl);
    else
        bin_reduce_kernel<0, true><<<n, kReduceBlock, 0, st>>>(offsets, grad_embeddings, L, wl, opt, loc, stl);
    NGP_CHECK_LAUNCH("grid_backward_binned_reduce_range");
    return NGP_OK;
}

extern "C" int ngp_x_grid_encode_backward_binned(const float *grad, const float *inputs, const int32_t *offsets,
                                                 float *grad_embeddings, const int32_t *B_dev, uint32_t B,
                                                 uint32_t grad_stride, uint32_t L, uint32_t max_level,
                                                 float S, uint32_t H, uint32_t gridtype, int align_corners,
                                                 uint32_t interp, uint32_t n_rows_total, uint32_t max_level_rows,
                                                 void *workspace, size_t workspace_bytes, ngp_stream_t stream)
{
    // (the tile-local layout needs the header reset only: stage 1)
    const int stage = ngp_x_grid_backward_binned_counts(B, L, n_rows_total, max_level_rows) ? 0 : 1;
    const int rc = ngp_x_grid_backward_binned_prepare(inputs, 0.0f, offsets, B_dev, B, L, max_level, S, H, gridtype,
                                                      align_corners, interp, n_rows_total, max_level_rows, 0, 0, stage, workspace,
                                                      workspace_bytes, stream);
    if (rc != NGP_OK) return rc;
    return ngp_x_grid_backward_binned_apply(grad, inputs, offsets, grad_embeddings, B_dev, B, grad_stride, L, max_level, S, H,
                                            gridtype, align_corners, interp, n_rows_total, max_level_rows, workspace,
                                            workspace_bytes, nullptr, nullptr, nullptr, nullptr, 0.0f, 0.0f, 0.0f, 0, nullptr, stream);
}
