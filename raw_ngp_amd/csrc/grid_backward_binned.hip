// Hash-grid table gradient without global float atomics (D = 3, C = 2: the field's encoder).
//
// Why: the reference's kernel_grid_backward (gridencoder/src/gridencoder.cu:252-349) issues one float
// atomicAdd per (sample, level, corner, channel).  On MI355X global float atomics execute memory-side
// at ~21 G requests/s when lanes hit different 64-B lines (measured: tools/ubench/atomics.hip), i.e.
// 3.3 ms for 2^18 samples -- 8x the whole step budget.  Plain stores and LDS atomics are ~50x faster,
// so the scatter is turned into "bin, then reduce in LDS":
//
//   plan    chunk = 4096 consecutive rows of one level (32 KiB of float2 accumulators); per-level chunk
//           bases come from `offsets` on the device
//   count   every (sample, level) computes its 8 corner rows; a per-workgroup LDS histogram over that
//           level's chunks is added to the global per-chunk counts (one atomic per workgroup and chunk)
//   scan    exclusive prefix of the counts -> record offsets; chunks heavier than kSeg records are
//           split into segments (coarse dense levels concentrate millions of records in a few chunks)
//   fill    same traversal; records {row-in-chunk, w*g.x, w*g.y} (12 B) are stored into their chunk's
//           range, positions handed out by LDS atomics + one global atomic per workgroup and chunk
//   reduce  one workgroup per (chunk, segment): stream the records (coalesced), ds_add into a 32 KiB
//           LDS image of the chunk, then grad_table[chunk] += image (plain read-modify-write when the
//           chunk has one segment, well-shaped contiguous atomics otherwise)
//
// Result: the same sums as the reference in a different (also unspecified) order.  HBM traffic is
// 2 x 12 B x 8 x L per sample of plain coalesced traffic instead of 16 x L scattered atomic requests.
#include "grid_common.hpp"

namespace ngp {

constexpr uint32_t kChunkRows = 4096;   // rows per chunk: 4096 x float2 = 32 KiB of LDS
constexpr uint32_t kChunkShift = 12;
constexpr uint32_t kSeg = 32768;        // records per reduce work item
constexpr uint32_t kMaxChunks = 12288;    // LDS histogram bound (48 KiB): tables up to 50 M rows

struct Record {
    uint32_t row;   // row inside the chunk
    float gx, gy;
};

// workspace header (uint32 words); arrays sized for n_chunks_max
struct WsLayout {
    uint32_t *chunk_base;   // [kMaxLevels + 1] first chunk of each level; [L] = total chunks
    uint32_t *count;        // [n_chunks_max]
    uint32_t *cursor;       // [n_chunks_max]
    uint32_t *offset;       // [n_chunks_max + 1] record offsets
    uint32_t *seg_base;     // [n_chunks_max + 1] first reduce work item of each chunk
    Record *records;
};

__host__ __device__ inline WsLayout ws_layout(void *ws, uint32_t n_chunks_max)
{
    WsLayout w;
    uint32_t *p = reinterpret_cast<uint32_t *>(ws);
    w.chunk_base = p;
    p += kMaxLevels + 1 + 3;   // keep 16-byte alignment below
    w.count = p;
    p += n_chunks_max;
    w.cursor = p;
    p += n_chunks_max;
    w.offset = p;
    p += n_chunks_max + 1;
    w.seg_base = p;
    p += n_chunks_max + 1;
    p += (4 - ((uintptr_t)(p - reinterpret_cast<uint32_t *>(ws)) & 3)) & 3;
    w.records = reinterpret_cast<Record *>(p);
    return w;
}

static inline size_t ws_bytes(uint32_t B, uint32_t L, uint32_t n_chunks_max)
{
    const size_t head = (size_t)(kMaxLevels + 4 + 4 * (size_t)n_chunks_max + 2 + 4) * 4;
    return head + (size_t)B * L * 8 * sizeof(Record) + 64;
}

// ------------------------------------------------------------------ plan
__global__ __launch_bounds__(1024) void bin_plan_kernel(const int32_t *__restrict__ offsets, uint32_t L,
                                                        uint32_t n_chunks_max, WsLayout w)
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
        total = run;
    }
    __syncthreads();
    const uint32_t n = min(total, n_chunks_max);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        w.count[i] = 0;
        w.cursor[i] = 0;
    }
}

// ------------------------------------------------------------------ count / fill
// One workgroup = 256 samples of one level.  FILL = false: histogram only.
template <bool FILL>
__global__ __launch_bounds__(kBlock) void bin_traverse_kernel(
    const float *__restrict__ grad, const float *__restrict__ inputs, const int32_t *__restrict__ offsets, uint32_t B,
    uint32_t ntiles, LevelRes lv, uint32_t gridtype, bool align_corners, uint32_t interp, WsLayout w)
{
    extern __shared__ uint32_t hist[];   // [bins of this level] counts, then (FILL) global bases
    const uint32_t item = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t level = item / ntiles;
    const uint32_t b = (item - level * ntiles) * kBlock + threadIdx.x;

    const Geom<3> g = make_geom<3>(offsets, level, lv.res[level], gridtype);
    const uint32_t first = w.chunk_base[level];
    const uint32_t nbins = w.chunk_base[level + 1] - first;
    for (uint32_t i = threadIdx.x; i < nbins; i += kBlock) hist[i] = 0;
    __syncthreads();

    bool live = b < B;
    Cell<3> cl;
    if (live) {
        float x[3];
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) x[d] = inputs[(size_t)b * 3 + d];
        live = locate<3>(x, g.res, align_corners, interp, cl);
    }
    uint32_t rows[8], pos[8];
    if (live) {
#pragma unroll
        for (uint32_t corner = 0; corner < 8; corner++) {
            uint32_t c[3];
#pragma unroll
            for (uint32_t d = 0; d < 3; d++) c[d] = (corner & (1u << d)) ? min(cl.c[d] + 1u, g.res - 1u) : cl.c[d];
            rows[corner] = row_of<3>(g, c);
            pos[corner] = atomicAdd(&hist[rows[corner] >> kChunkShift], 1u);
        }
    }
    __syncthreads();
    if (!FILL) {
        for (uint32_t i = threadIdx.x; i < nbins; i += kBlock)
            if (hist[i]) atomicAdd(&w.count[first + i], hist[i]);
        return;
    }
    // reserve this workgroup's range in every chunk it touches; hist[i] becomes the global base
    for (uint32_t i = threadIdx.x; i < nbins; i += kBlock) {
        const uint32_t n = hist[i];
        hist[i] = n ? w.offset[first + i] + atomicAdd(&w.cursor[first + i], n) : 0u;
    }
    __syncthreads();
    if (!live) return;
    const float2 gr = reinterpret_cast<const float2 *>(grad)[(size_t)level * B + b];
#pragma unroll
    for (uint32_t corner = 0; corner < 8; corner++) {
        float wgt = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) wgt *= (corner & (1u << d)) ? cl.f[d] : 1.0f - cl.f[d];
        Record r;
        r.row = rows[corner] & (kChunkRows - 1u);
        r.gx = wgt * gr.x;
        r.gy = wgt * gr.y;
        w.records[hist[rows[corner] >> kChunkShift] + pos[corner]] = r;
    }
}

// ------------------------------------------------------------------ scan
__global__ __launch_bounds__(1024) void bin_scan_kernel(uint32_t L, WsLayout w)
{
    __shared__ uint32_t wave_a[16], wave_b[16];
    __shared__ uint32_t carry_a, carry_b;
    const uint32_t n = w.chunk_base[L];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (tid == 0) carry_a = carry_b = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024u) {
        const uint32_t i = base + tid;
        const uint32_t cnt = i < n ? w.count[i] : 0u;
        const uint32_t seg = i < n ? max(1u, (cnt + kSeg - 1) / kSeg) : 0u;
        uint32_t a = cnt, s = seg;
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
        for (uint32_t k = 0; k < 16u; k++) {
            if (k < wid) {
                oa += wave_a[k];
                ob += wave_b[k];
            }
            ta += wave_a[k];
            tb += wave_b[k];
        }
        const uint32_t ca = carry_a, cb = carry_b;
        if (i < n) {
            w.offset[i] = ca + oa + a - cnt;
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

// ------------------------------------------------------------------ reduce
__global__ __launch_bounds__(kBlock) void bin_reduce_kernel(const int32_t *__restrict__ offsets,
                                                           float *__restrict__ grad_table, uint32_t L, WsLayout w)
{
    __shared__ float acc[kChunkRows * 2];
    __shared__ uint32_t s_chunk;
    const uint32_t n_chunks = w.chunk_base[L];
    const uint32_t n_items = w.seg_base[n_chunks];
    const uint32_t item = blockIdx.x;
    if (item >= n_items) return;
    if (threadIdx.x == 0) {   // largest chunk with seg_base[chunk] <= item
        uint32_t lo = 0, hi = n_chunks;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (w.seg_base[mid] <= item)
                lo = mid;
            else
                hi = mid;
        }
        s_chunk = lo;
    }
    for (uint32_t i = threadIdx.x; i < kChunkRows * 2; i += kBlock) acc[i] = 0.0f;
    __syncthreads();
    const uint32_t chunk = s_chunk;
    const uint32_t seg = item - w.seg_base[chunk];
    const uint32_t n_seg = w.seg_base[chunk + 1] - w.seg_base[chunk];
    const uint32_t beg = w.offset[chunk] + seg * kSeg;
    const uint32_t end = min(w.offset[chunk + 1], beg + kSeg);
    if (beg >= end) return;   // empty chunk: nothing to add

    const uint32_t *rec = reinterpret_cast<const uint32_t *>(w.records);
    for (uint32_t i = beg + threadIdx.x; i < end; i += kBlock) {
        const uint32_t row = __builtin_nontemporal_load(rec + (size_t)i * 3);
        const float gx = __uint_as_float(__builtin_nontemporal_load(rec + (size_t)i * 3 + 1));
        const float gy = __uint_as_float(__builtin_nontemporal_load(rec + (size_t)i * 3 + 2));
        atomicAdd(&acc[row * 2], gx);
        atomicAdd(&acc[row * 2 + 1], gy);
    }
    __syncthreads();

    // which level does this chunk belong to, and where does it start in the table
    uint32_t level = 0;
    while (level + 1 < L && w.chunk_base[level + 1] <= chunk) level++;
    const uint32_t row0 = (chunk - w.chunk_base[level]) << kChunkShift;
    const uint32_t T = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t rows_here = min(kChunkRows, T - row0);
    float *dst = grad_table + ((size_t)(uint32_t)offsets[level] + row0) * 2;
    if (n_seg == 1) {
        float2 *d2 = reinterpret_cast<float2 *>(dst);
        const float2 *a2 = reinterpret_cast<const float2 *>(acc);
        for (uint32_t i = threadIdx.x; i < rows_here; i += kBlock) {
            float2 v = d2[i];
            v.x += a2[i].x;
            v.y += a2[i].y;
            d2[i] = v;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < rows_here * 2; i += kBlock)
            if (acc[i] != 0.0f) unsafeAtomicAdd(dst + i, acc[i]);
    }
}

}  // namespace ngp

using namespace ngp;

extern "C" size_t ngp_x_grid_backward_workspace_bytes(uint32_t B, uint32_t L, uint32_t n_rows_total)
{
    const uint32_t n_chunks_max = n_rows_total / kChunkRows + L + 1;
    return ws_bytes(B, L, n_chunks_max);
}

extern "C" int ngp_x_grid_encode_backward_binned(const float *grad, const float *inputs, const int32_t *offsets,
                                                 float *grad_embeddings, uint32_t B, uint32_t L, uint32_t max_level,
                                                 float S, uint32_t H, uint32_t gridtype, int align_corners,
                                                 uint32_t interp, uint32_t n_rows_total, void *workspace,
                                                 size_t workspace_bytes, ngp_stream_t stream)
{
    if (B == 0 || max_level == 0) return NGP_OK;
    NGP_REQUIRE(grad && inputs && offsets && grad_embeddings && workspace, "grid_encode_backward_binned: null tensor");
    LevelRes lv;
    NGP_REQUIRE(fill_levels(lv, S, H, L), "grid_encode_backward_binned: L must be in [1, %u]", kMaxLevels);
    NGP_REQUIRE(max_level <= L, "grid_encode_backward_binned: max_level > L");
    NGP_REQUIRE(((uintptr_t)workspace & 15u) == 0, "grid_encode_backward_binned: workspace must be 16-byte aligned");
    const uint32_t n_chunks_max = n_rows_total / kChunkRows + L + 1;
    NGP_REQUIRE(workspace_bytes >= ws_bytes(B, L, n_chunks_max), "grid_encode_backward_binned: workspace too small");
    NGP_REQUIRE((uint64_t)B * L * 8 < (1ull << 32), "grid_encode_backward_binned: B * L too large");
    // the per-workgroup LDS histogram must hold the chunks of the largest level
    NGP_REQUIRE(n_chunks_max <= kMaxChunks, "grid_encode_backward_binned: table too large (> %u chunks)", kMaxChunks);
    hipStream_t st = as_stream(stream);
    const WsLayout w = ws_layout(workspace, n_chunks_max);
    const uint32_t ntiles = ceil_div(B, kBlock);
    const size_t hist_bytes = (size_t)n_chunks_max * 4;

    bin_plan_kernel<<<1, 1024, 0, st>>>(offsets, L, n_chunks_max, w);
    bin_traverse_kernel<false><<<ntiles * max_level, kBlock, hist_bytes, st>>>(grad, inputs, offsets, B, ntiles, lv,
                                                                               gridtype, align_corners != 0, interp, w);
    bin_scan_kernel<<<1, 1024, 0, st>>>(L, w);
    bin_traverse_kernel<true><<<ntiles * max_level, kBlock, hist_bytes, st>>>(grad, inputs, offsets, B, ntiles, lv,
                                                                              gridtype, align_corners != 0, interp, w);
    const uint32_t n_items_max = n_chunks_max + (uint32_t)(((uint64_t)B * max_level * 8) / kSeg) + 1;
    bin_reduce_kernel<<<n_items_max, kBlock, 0, st>>>(offsets, grad_embeddings, L, w);
    NGP_CHECK_LAUNCH("grid_encode_backward_binned");
    return NGP_OK;
}
