// Density-grid refresh on the device (extension; the reference does this with ~60 small torch kernels and three
// host round trips, NeRFRenderer.update_extra_state, nerf/renderer.py:811-897).
//
//   sample    cells to re-evaluate for one cascade: n_uniform cells drawn uniformly (or, in `full` mode, every cell
//             once) followed by n_occupied cells drawn uniformly among the cells with density > 0
//             (renderer.py:851-866), each with a position jittered inside the cell (:868-872)
//   (caller)  density at those positions: hash-grid encode + density MLP
//   scatter   tmp[cell] = sigma (renderer.py:881; duplicates resolve to the larger value instead of "last writer")
//   update    grid = max(grid * decay, tmp) where both are >= 0 (:884-885); sum of clamp(grid, 0) for the mean (:887)
//   packbits  with thresh = min(mean, density_thresh) read from device memory (:890-894)
//
// "uniformly among the occupied cells" without torch.nonzero: a 64-cell occupancy mask per word + an exclusive prefix
// of the popcounts; a draw r in [0, n_pos) is located by binary search over the prefix and a select inside the word.
#include "morton.hpp"
#include "rng_common.hpp"

namespace ngp {

// workspace: mask[n_words] u64 | prefix[n_words + 1] u32
struct GridWs {
    uint64_t *mask;
    uint32_t *prefix;
};
__host__ __device__ inline GridWs grid_ws(void *ws, uint32_t n_words)
{
    GridWs g;
    g.mask = reinterpret_cast<uint64_t *>(ws);
    g.prefix = reinterpret_cast<uint32_t *>(g.mask + n_words);
    return g;
}

// one wave per 64 cells: coalesced read, ballot -> mask word; prefix[w + 1] holds the word's popcount until the scan
__global__ __launch_bounds__(256) void grid_positive_mask_kernel(const float *__restrict__ grid, uint32_t n_words,
                                                                GridWs g)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
    for (uint32_t w = wave; w < n_words; w += n_waves) {
        const unsigned long long m = __ballot(grid[(size_t)w * 64 + lane] > 0.0f);
        if (lane == 0) {
            g.mask[w] = m;
            g.prefix[w + 1] = (uint32_t)__popcll(m);
        }
    }
}

// in place: prefix[w + 1] = counts -> prefix[w] = cells before word w, prefix[n_words] = total.
// One workgroup; wave q owns a contiguous segment and walks it 64 words at a time (coalesced), 32 rows of loads in
// flight before any store (a per-lane range of consecutive words would make every load a different cache line).
__global__ __launch_bounds__(1024) void grid_prefix_kernel(uint32_t n_words, GridWs g)
{
    __shared__ uint32_t wave_sum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t seg = ((n_words + 15u) / 16u + 63u) & ~63u;
    const uint32_t lo = min(n_words, wid * seg), hi = min(n_words, lo + seg);
    uint32_t mine = 0;
    for (uint32_t w = lo + lane; w < hi; w += 64u) mine += g.prefix[w + 1];
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (lane == 0) wave_sum[wid] = mine;
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t k = 0; k < wid; k++) run += wave_sum[k];
    if (tid == 0) g.prefix[0] = 0;
    for (uint32_t base = lo; base < hi; base += 32u * 64u) {
        uint32_t c[32];
#pragma unroll
        for (uint32_t r = 0; r < 32; r++) {
            const uint32_t w = base + r * 64u + lane;
            c[r] = w < hi ? g.prefix[w + 1] : 0u;
        }
#pragma unroll
        for (uint32_t r = 0; r < 32; r++) {
            uint32_t v = c[r];
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                const uint32_t up = __shfl_up(v, d, 64);
                if (lane >= d) v += up;
            }
            const uint32_t w = base + r * 64u + lane;
            if (w < hi) g.prefix[w + 1] = run + v;
            run += __shfl(v, 63, 64);
        }
    }
}

// index of the r-th (0-based) set bit of m
__device__ __forceinline__ uint32_t select_bit(uint64_t m, uint32_t r)
{
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t width = 32; width >= 1; width >>= 1) {
        const uint64_t low = m & ((1ull << width) - 1ull);
        const uint32_t c = (uint32_t)__popcll(low);
        if (r >= c) {
            r -= c;
            m >>= width;
            pos += width;
        } else {
            m = low;
        }
    }
    return pos;
}

// cell of draw i (Morton index; -1: the occupied half when nothing is occupied yet) and its grid coordinates
__device__ __forceinline__ int32_t draw_cell(const GridWs &g, uint32_t n_words, uint32_t H, uint32_t n_uniform, bool full,
                                             uint32_t i, const uint32_t r[4], uint32_t &cx, uint32_t &cy, uint32_t &cz)
{
    if (i < n_uniform) {
        if (full) {
            cx = compact_bits(i);
            cy = compact_bits(i >> 1);
            cz = compact_bits(i >> 2);
            return (int32_t)i;
        }
        cx = __umulhi(r[0], H);
        cy = __umulhi(r[1], H);
        cz = __umulhi(r[2], H);
        return (int32_t)morton3(cx, cy, cz);
    }
    const uint32_t n_pos = g.prefix[n_words];
    if (n_pos == 0) {   // nothing occupied yet: the reference leaves this half out
        cx = cy = cz = 0;
        return -1;
    }
    const uint32_t pick = __umulhi(r[0], n_pos);
    uint32_t lo = 0, hi = n_words;   // largest w with prefix[w] <= pick
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g.prefix[mid] <= pick)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t cell = lo * 64u + select_bit(g.mask[lo], pick - g.prefix[lo]);
    cx = compact_bits(cell);
    cy = compact_bits(cell >> 1);
    cz = compact_bits(cell >> 2);
    return (int32_t)cell;
}

// position inside the cell: (2 c / (H - 1) - 1) * (bound - half) + (2 u - 1) * half      (renderer.py:868-872)
__device__ __forceinline__ void store_draw(uint32_t slot, int32_t index, uint32_t cx, uint32_t cy, uint32_t cz, uint32_t H,
                                           float span, float half, const uint32_t q[4], int32_t *__restrict__ indices,
                                           float *__restrict__ xyzs)
{
    indices[slot] = index;
    if (index < 0) {
        xyzs[(size_t)slot * 3] = xyzs[(size_t)slot * 3 + 1] = xyzs[(size_t)slot * 3 + 2] = 0.0f;
        return;
    }
    const float inv = (float)(H - 1u);
    // one 12-byte store (global_store_dwordx3) instead of three: the slots of a sorted draw are scattered
    struct P3 {
        float x, y, z;
    };
    P3 p;
    p.x = ((2.0f * (float)cx) / inv - 1.0f) * span + (u01(q[0]) * 2.0f - 1.0f) * half;
    p.y = ((2.0f * (float)cy) / inv - 1.0f) * span + (u01(q[1]) * 2.0f - 1.0f) * half;
    p.z = ((2.0f * (float)cz) / inv - 1.0f) * span + (u01(q[2]) * 2.0f - 1.0f) * half;
    reinterpret_cast<P3 *>(xyzs)[slot] = p;
}

__global__ __launch_bounds__(256) void grid_sample_cells_kernel(GridWs g, uint32_t n_words, uint32_t H, float span,
                                                               float half, uint32_t n_uniform, uint32_t n_occupied,
                                                               bool full, uint32_t seed_lo, uint32_t seed_hi,
                                                               const uint32_t *__restrict__ draw_dev, uint32_t draw,
                                                               int32_t *__restrict__ indices, float *__restrict__ xyzs)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_uniform + n_occupied) return;
    if (draw_dev) draw = draw_dev[0];
    uint32_t r[4] = {i, draw, 2u, 0u}, q[4] = {i, draw, 3u, 0u};
    philox4x32_10(r, seed_lo, seed_hi);
    philox4x32_10(q, seed_lo, seed_hi);
    uint32_t cx, cy, cz;
    const int32_t index = draw_cell(g, n_words, H, n_uniform, full, i, r, cx, cy, cz);
    store_draw(i, index, cx, cy, cz, H, span, half, q, indices, xyzs);
}

// ---- random draws that are BORN in Morton order ---------------------------------------------------------------------
// The cells are evaluated by the hash-grid encoder, whose coarse and middle levels hit the caches only if neighbouring
// points arrive together: 2^19 uniformly drawn cells in draw order cost the encoder 225 us, the same cells sorted 183 (and
// the scatter of their densities 25 us instead of 44).  Rounds 3 and 4 SORTED the draws (counting sort on a 12-bit key:
// first with one global atomic per draw, then with LDS atomics and a place pass of scattered 16-byte stores -- 101 us on
// the side stream, beside a step whose MLP backward it slowed by ~ 35 us).  There is nothing to sort if the draws are
// generated bin by bin.  n independent uniform draws are, exactly: (1) a multinomial count per bin, (2) inside every bin,
// that many independent uniform draws from the bin.  So:
//   count   (1) is taken from the draws themselves: workgroup w histograms the keys of draws [w T, (w + 1) T) of the
//           counter-based stream (i, draw, 2) in LDS and writes row w of wg_hist[workgroups][4096]; no atomics leave the CU
//   totals  column sums of the rows -> hist[half][bin]
//   scan    exclusive scan -> base[bin]: the first output slot of every bin
//   emit    (2): one lane per OUTPUT SLOT j.  Its bin is the one whose [base, next base) holds j; its position inside the
//           bin comes from a fresh number of the stream (j, draw, 4); the jitter inside the cell from (j, draw, 5).
//           Coalesced stores, no cursor, no atomics; the order inside a bin is the slot order -- reproducible.
// Keys, 4096 bins per half, uniform over the bins:
//   uniform half   the top 12 bits of the cell's Morton index; inside a bin the low bits of the index are uniform (H a
//                  power of two: every cell equally likely) -> cell = bin << shift | fresh bits
//   occupied half  the top 12 bits of the draw's random number u: the pick floor(u n_pos / 2^32) enumerates the occupied
//                  cells in Morton order, so sorting by u sorts by cell; inside a bin the low 20 bits of u are uniform
//                  -> u = bin << 20 | fresh bits, then the same pick and search as an unsorted draw
// The multiset of cells has the distribution of the reference's independent draws (renderer.py:851-866); it is not the
// multiset the unsorted kernel above draws from the same seed (oracle: density_grid_sample(..., binned=True)).
constexpr uint32_t kSortBins = 8192;   // 4096 per half
constexpr uint32_t kSortTile = 2048;   // draws per counting workgroup (a divisor of each half's size, or the launcher falls back)
constexpr uint32_t kSortBlock = 512;   // ... of this many lanes
constexpr uint32_t kSortPer = kSortTile / kSortBlock;
constexpr uint32_t kSortHalf = kSortBins / 2u;
struct SortWs {
    uint32_t *hist, *base;             // kSortBins words each: bin totals, first slot of every bin
    uint32_t *wg_hist;                 // [workgroups][kSortHalf]: counts
};
__host__ __device__ inline SortWs sort_ws(void *ws, uint32_t n_words)
{
    SortWs s;
    s.hist = grid_ws(ws, n_words).prefix + n_words + 1;
    s.base = s.hist + kSortBins;
    s.wg_hist = s.base + kSortBins;
    return s;
}
// 12-bit key inside the draw's half
__device__ __forceinline__ uint32_t sort_key(uint32_t i, uint32_t n_uniform, uint32_t H, uint32_t shift, const uint32_t r[4])
{
    if (i < n_uniform) return morton3(__umulhi(r[0], H), __umulhi(r[1], H), __umulhi(r[2], H)) >> shift;
    return r[0] >> 20;
}

__global__ __launch_bounds__(kSortBlock) void grid_sort_count_kernel(SortWs s, uint32_t H, uint32_t n_uniform, uint32_t n,
                                                                    uint32_t shift, uint32_t seed_lo, uint32_t seed_hi,
                                                                    const uint32_t *__restrict__ draw_dev, uint32_t draw)
{
    __shared__ uint32_t h[kSortHalf];
    for (uint32_t k = threadIdx.x; k < kSortHalf; k += kSortBlock) h[k] = 0u;
    __syncthreads();
    if (draw_dev) draw = draw_dev[0];
    const uint32_t i0 = blockIdx.x * kSortTile;
#pragma unroll
    for (uint32_t j = 0; j < kSortPer; j++) {
        const uint32_t i = i0 + j * kSortBlock + threadIdx.x;
        if (i < n) {
            uint32_t r[4] = {i, draw, 2u, 0u};
            philox4x32_10(r, seed_lo, seed_hi);
            atomicAdd(&h[sort_key(i, n_uniform, H, shift, r)], 1u);
        }
    }
    __syncthreads();
    uint32_t *row = s.wg_hist + (size_t)blockIdx.x * kSortHalf;
    for (uint32_t k = threadIdx.x; k < kSortHalf; k += kSortBlock) row[k] = h[k];
}

// Column sums of wg_hist over a half's workgroups: 32 bins x 8 row segments per workgroup of 256 lanes.  A lane takes its
// segment's rows in batches of 16 independent loads (consecutive lanes = consecutive bins: 128-byte rows), the eight segment
// totals of a bin meet in LDS.  (One lane per column, walking it row by row, took 98 us: 256 dependent round trips.)
__global__ __launch_bounds__(256) void grid_sort_totals_kernel(SortWs s, uint32_t wg_uniform, uint32_t wg_total)
{
    __shared__ uint32_t seg_sum[8][32];
    const uint32_t b = threadIdx.x & 31u, sg = threadIdx.x >> 5;
    const uint32_t t = blockIdx.x * 32u + b;                       // (half, bin); kSortBins is a multiple of 32
    const uint32_t half = t / kSortHalf, bin = t - half * kSortHalf;
    const uint32_t w0 = half ? wg_uniform : 0u, w1 = half ? wg_total : wg_uniform;
    const uint32_t per = (w1 - w0 + 7u) / 8u, r0 = min(w1, w0 + sg * per), r1 = min(w1, r0 + per);
    const uint32_t *col = s.wg_hist + bin;
    uint32_t sum = 0;
    for (uint32_t w = r0; w < r1; w += 16u) {
        uint32_t c[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) c[k] = w + k < r1 ? col[(size_t)(w + k) * kSortHalf] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) sum += c[k];
    }
    seg_sum[sg][b] = sum;
    __syncthreads();
    if (sg == 0) {
        uint32_t total = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) total += seg_sum[k][b];
        s.hist[t] = total;
    }
}

__global__ __launch_bounds__(1024) void grid_sort_scan_kernel(SortWs s)   // one workgroup: exclusive scan of the counts
{
    constexpr uint32_t K = kSortBins / 1024u;
    __shared__ uint32_t wave_sum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    uint32_t c[K], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < K; k++) {
        c[k] = s.hist[tid * K + k];
        sum += c[k];
    }
    uint32_t inc = sum;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    if (lane == 63u) wave_sum[wid] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (uint32_t k = 0; k < wid; k++) run += wave_sum[k];
#pragma unroll
    for (uint32_t k = 0; k < K; k++) {
        s.base[tid * K + k] = run;
        run += c[k];
    }
}

// one lane per output slot (see above)
__global__ __launch_bounds__(256) void grid_sort_emit_kernel(GridWs g, SortWs s, uint32_t n_words, uint32_t H, float span,
                                                            float half, uint32_t n_uniform, uint32_t n_occupied,
                                                            uint32_t shift, uint32_t seed_lo, uint32_t seed_hi,
                                                            const uint32_t *__restrict__ draw_dev, uint32_t draw,
                                                            int32_t *__restrict__ indices, float *__restrict__ xyzs)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x, n = n_uniform + n_occupied;
    if (j >= n) return;
    if (draw_dev) draw = draw_dev[0];
    const bool occupied = j >= n_uniform;       // (base[kSortHalf] == n_uniform: the halves stay where they were)
    const uint32_t *base = s.base + (occupied ? kSortHalf : 0u);
    uint32_t lo = 0, hi = kSortHalf;            // largest bin with base[bin] <= j (an empty bin shares its base with the next)
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (base[mid] <= j)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t bin = lo;
    uint32_t r[4] = {j, draw, 4u, 0u}, q[4] = {j, draw, 5u, 0u};
    philox4x32_10(r, seed_lo, seed_hi);
    philox4x32_10(q, seed_lo, seed_hi);
    uint32_t cell;
    int32_t index;
    if (!occupied) {
        cell = (bin << shift) | (shift ? r[0] >> (32u - shift) : 0u);
        index = (int32_t)cell;
    } else {
        const uint32_t n_pos = g.prefix[n_words];
        if (n_pos == 0u) {                      // nothing occupied yet: the reference leaves this half out
            cell = 0;
            index = -1;
        } else {
            const uint32_t u = (bin << 20) | (r[0] >> 12), pick = __umulhi(u, n_pos);
            uint32_t wl = 0, wh = n_words;      // largest w with prefix[w] <= pick (as draw_cell)
            while (wh - wl > 1u) {
                const uint32_t mid = (wl + wh) >> 1;
                if (g.prefix[mid] <= pick)
                    wl = mid;
                else
                    wh = mid;
            }
            cell = wl * 64u + select_bit(g.mask[wl], pick - g.prefix[wl]);
            index = (int32_t)cell;
        }
    }
    store_draw(j, index, compact_bits(cell), compact_bits(cell >> 1), compact_bits(cell >> 2), H, span, half, q, indices, xyzs);
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const int32_t *__restrict__ indices,
                                                          const float *__restrict__ sigmas, uint32_t n,
                                                          float *__restrict__ tmp)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int32_t idx = indices[i];
    if (idx < 0) return;
    // sigma >= 0, tmp starts at -1: as signed integers, float bit patterns of non-negative values order like the floats
    atomicMax(reinterpret_cast<int *>(tmp) + idx, __float_as_int(sigmas[i]));
}

constexpr uint32_t kUpdateBlocks = 1024;   // partial sums of the density mean: stats[4 .. 4 + kUpdateBlocks)

__global__ __launch_bounds__(256) void grid_update_kernel(float *__restrict__ grid, float *__restrict__ tmp, uint32_t n4,
                                                         float decay, float *__restrict__ stats)
{
    __shared__ float wave_sum[4];
    float sum = 0.0f;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        float4 v = reinterpret_cast<float4 *>(grid)[i];
        const float4 t = reinterpret_cast<float4 *>(tmp)[i];
#define NGP_UPD(c)                                                  \
    if (v.c >= 0.0f && t.c >= 0.0f) v.c = fmaxf(v.c * decay, t.c);  \
    sum += fmaxf(v.c, 0.0f);
        NGP_UPD(x) NGP_UPD(y) NGP_UPD(z) NGP_UPD(w)
#undef NGP_UPD
        reinterpret_cast<float4 *>(grid)[i] = v;
        reinterpret_cast<float4 *>(tmp)[i] = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);   // ready for the next refresh
    }
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63u) == 0) wave_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    // one partial per block, added up in block order by the packbits kernel: no float atomics, so the threshold -- and
    // with it the whole training run -- does not depend on the order in which blocks finish
    if (threadIdx.x == 0) stats[4 + blockIdx.x] = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
}

// packbits (raymarching.cu:267-289) with thresh = min(mean density, density_thresh) taken from the device
__global__ __launch_bounds__(256) void packbits_mean_kernel(const float *__restrict__ grid, uint32_t N, float *stats,
                                                           float inv_cells, float density_thresh,
                                                           uint8_t *__restrict__ bitfield)
{
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    __shared__ float s_total;
    if (threadIdx.x < 64u) {   // fixed summation order: lane l adds partials l, l + 64, ...; then a butterfly
        float acc = 0.0f;
        for (uint32_t i = threadIdx.x; i < kUpdateBlocks; i += 64u) acc += stats[4 + i];
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (threadIdx.x == 0) s_total = acc;
    }
    __syncthreads();
    const float mean = s_total * inv_cells;
    const float thresh = fminf(mean, density_thresh);
    if (n == 0) {
        stats[0] = s_total;
        stats[1] = mean;
        stats[2] = thresh;
    }
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2];
    const float4 b = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2 + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;
    bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;
    bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;
    bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;
    bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

}  // namespace ngp

using namespace ngp;

extern "C" size_t ngp_x_density_grid_workspace_bytes(uint32_t H)
{
    const size_t n_words = (size_t)H * H * H / 64;
    // mask | prefix | sort: column totals, bases, one histogram row per workgroup of the two halves' draws (H^3 / 2 of them)
    const size_t sort_wgs = ((size_t)H * H * H / 2 + kSortTile - 1) / kSortTile + 2;
    return n_words * 8 + (n_words + 1) * 4 + 2 * (size_t)kSortBins * 4 + sort_wgs * kSortHalf * 4 + 64;
}

extern "C" int ngp_x_density_grid_sample(const float *grid_cas, uint32_t H, float span, float half, uint32_t n_uniform,
                                         uint32_t n_occupied, int full, uint64_t seed, const uint32_t *draw_dev,
                                         uint32_t draw, void *workspace, size_t workspace_bytes, int32_t *indices,
                                         float *xyzs, ngp_stream_t stream)
{
    NGP_REQUIRE(grid_cas && workspace && indices && xyzs, "density_grid_sample: null tensor");
    const uint64_t cells = (uint64_t)H * H * H;
    NGP_REQUIRE(H >= 4 && H <= 1024 && cells % 64 == 0, "density_grid_sample: H^3 must be a multiple of 64, H <= 1024");
    NGP_REQUIRE(!full || n_uniform == cells, "density_grid_sample: a full sweep takes n_uniform = H^3");
    NGP_REQUIRE((uint64_t)n_uniform + n_occupied < (1ull << 31), "density_grid_sample: too many cells");
    NGP_REQUIRE(workspace_bytes >= ngp_x_density_grid_workspace_bytes(H) && ((uintptr_t)workspace & 7u) == 0,
                "density_grid_sample: workspace too small or misaligned");
    const uint32_t n_words = (uint32_t)(cells / 64), n = n_uniform + n_occupied;
    if (n == 0) return NGP_OK;
    hipStream_t st = as_stream(stream);
    const GridWs g = grid_ws(workspace, n_words);
    if (n_occupied) {
        grid_positive_mask_kernel<<<dim3(min(ceil_div(n_words, 4u), 2048u)), dim3(256), 0, st>>>(grid_cas, n_words, g);
        grid_prefix_kernel<<<dim3(1), dim3(1024), 0, st>>>(n_words, g);
    }
    // random draws leave in Morton order of their cells (bins of cells ascending): what the encoder that evaluates them
    // wants.  NGP_REFRESH_SORT=0: independent draws in draw order (a full sweep is in Morton order as it is)
    static const bool sort_on = !(getenv("NGP_REFRESH_SORT") && getenv("NGP_REFRESH_SORT")[0] == '0');
    // (the binned draw wants a power-of-two grid of at least 4096 cells, whole counting workgroups of one half and a workspace
    // row for each of them: anything else is drawn unsorted)
    uint32_t bits = 0;
    while ((1ull << bits) < cells) bits++;
    const uint32_t wgs = ceil_div(n, kSortTile), wg_uniform = n_uniform / kSortTile;
    const bool sortable = (H & (H - 1u)) == 0 && bits >= 12u && n_uniform % kSortTile == 0 &&
                          (size_t)wgs <= ((size_t)cells / 2 + kSortTile - 1) / kSortTile + 2;
    if (full || !sort_on || !sortable) {
        grid_sample_cells_kernel<<<dim3(ceil_div(n, 256u)), dim3(256), 0, st>>>(
            g, n_words, H, span, half, n_uniform, n_occupied, full != 0, (uint32_t)seed, (uint32_t)(seed >> 32), draw_dev, draw,
            indices, xyzs);
    } else {
        const SortWs s = sort_ws(workspace, n_words);
        const uint32_t shift = bits - 12u;
        grid_sort_count_kernel<<<dim3(wgs), dim3(kSortBlock), 0, st>>>(s, H, n_uniform, n, shift, (uint32_t)seed,
                                                               (uint32_t)(seed >> 32), draw_dev, draw);
        grid_sort_totals_kernel<<<dim3(kSortBins / 32u), dim3(256), 0, st>>>(s, wg_uniform, wgs);
        grid_sort_scan_kernel<<<dim3(1), dim3(1024), 0, st>>>(s);
        grid_sort_emit_kernel<<<dim3(ceil_div(n, 256u)), dim3(256), 0, st>>>(g, s, n_words, H, span, half, n_uniform, n_occupied,
                                                                        shift, (uint32_t)seed, (uint32_t)(seed >> 32), draw_dev,
                                                                        draw, indices, xyzs);
    }
    NGP_CHECK_LAUNCH("density_grid_sample");
    return NGP_OK;
}

extern "C" int ngp_x_density_grid_scatter(const int32_t *indices, const float *sigmas, uint32_t n, float *tmp_cas,
                                          ngp_stream_t stream)
{
    if (n == 0) return NGP_OK;
    NGP_REQUIRE(indices && sigmas && tmp_cas, "density_grid_scatter: null tensor");
    grid_scatter_kernel<<<dim3(ceil_div(n, 256u)), dim3(256), 0, as_stream(stream)>>>(indices, sigmas, n, tmp_cas);
    NGP_CHECK_LAUNCH("density_grid_scatter");
    return NGP_OK;
}

extern "C" int ngp_x_density_grid_update(float *grid, float *tmp, uint32_t n_cells, float decay, float *stats,
                                         ngp_stream_t stream)
{
    NGP_REQUIRE(grid && tmp && stats, "density_grid_update: null tensor");
    NGP_REQUIRE(n_cells % 4 == 0 && (((uintptr_t)grid | (uintptr_t)tmp) & 15u) == 0,
                "density_grid_update: grids must be 16-byte aligned with a multiple of 4 cells");
    hipStream_t st = as_stream(stream);
    // (no hipMemsetAsync here: as a memset NODE of a captured graph it was seen to run after the update kernel under
    // rocprofv3, leaving a zero mean -> threshold 0 -> an all-occupied bitfield.  Every block writes its own partial.)
    grid_update_kernel<<<dim3(kUpdateBlocks), dim3(256), 0, st>>>(grid, tmp, n_cells / 4, decay, stats);
    NGP_CHECK_LAUNCH("density_grid_update");
    return NGP_OK;
}

extern "C" int ngp_x_packbits_mean(const float *grid, uint32_t N, float *stats, float density_thresh, uint8_t *bitfield,
                                   ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(grid && stats && bitfield, "packbits_mean: null tensor");
    NGP_REQUIRE(((uintptr_t)grid & 15u) == 0, "packbits_mean: grid must be 16-byte aligned");
    packbits_mean_kernel<<<dim3(ceil_div(N, 256u)), dim3(256), 0, as_stream(stream)>>>(grid, N, stats, 1.0f / (8.0f * (float)N),
                                                                                      density_thresh, bitfield);
    NGP_CHECK_LAUNCH("packbits_mean");
    return NGP_OK;
}
