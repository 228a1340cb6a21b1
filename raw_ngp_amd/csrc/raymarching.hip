// Density-grid guided ray marcher and volumetric compositor for gfx950.
//
// Behaviour (reference tree): raymarching/src/raymarching.cu
//   :42-81 mip / Morton helpers     :91-145 near_far_from_aabb   :162-198 sph_from_ray
//   :214-254 morton3D(_invert)      :267-289 packbits            :303-319 flatten_rays
//   :337-491 march_rays_train       :519-597 / :623-712 composite_rays_train fwd / bwd
//   :731-846 march_rays             :860-941 composite_rays
// Differences in design: sample offsets come from a wave-ballot / prefix-sum scan in ray
// order (deterministic; the reference's atomicAdd order is not), the arena variant marches
// ONCE (sample start times go to a scratch slab) and expands samples with one lane per
// sample and coalesced stores, flatten_rays is sample-parallel.  The stepping arithmetic is
// written operation for operation like oracle/ngp_oracle.c so that per-ray sample counts
// agree bit for bit.
#include "ngp_common.hpp"
#include "morton.hpp"

namespace ngp {

constexpr uint32_t kRayBlock = 64;   // one wave per workgroup: divergent per-ray loops
constexpr uint32_t kFlatBlock = 256;
constexpr float kSqrt3 = 1.7320508075688772f;
constexpr float kRPi = 0.3183098861837907f;

// frexp exponent clamped to [0, cascades-1]
__device__ __forceinline__ int mip_of(float mx, float cascades)
{
    int e;
    frexpf(mx, &e);
    return (int)fminf(cascades - 1.0f, fmaxf(0.0f, (float)e));
}

// ------------------------------------------------------------------ small per-ray / per-cell kernels
__global__ void near_far_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                const float *__restrict__ aabb, uint32_t N, float min_near,
                                float *__restrict__ nears, float *__restrict__ fars)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float tn = 0.0f, tf = 0.0f;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float o = rays_o[(size_t)n * 3 + a], r = 1.0f / rays_d[(size_t)n * 3 + a];
        float t0 = (aabb[a] - o) * r, t1 = (aabb[3 + a] - o) * r;
        if (t0 > t1) {
            const float s = t0;
            t0 = t1;
            t1 = s;
        }
        if (a == 0) {
            tn = t0;
            tf = t1;
        } else if (!miss) {
            if (tn > t1 || t0 > tf) {
                miss = true;
            } else {
                if (t0 > tn) tn = t0;
                if (t1 < tf) tf = t1;
            }
        }
    }
    if (miss) {
        nears[n] = fars[n] = 3.402823466e+38f;
        return;
    }
    if (tn < min_near) tn = min_near;
    nears[n] = tn;
    fars[n] = tf;
}

__global__ void sph_from_ray_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float radius,
                                    uint32_t N, float *__restrict__ coords)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[(size_t)n * 3], oy = rays_o[(size_t)n * 3 + 1], oz = rays_o[(size_t)n * 3 + 2];
    const float dx = rays_d[(size_t)n * 3], dy = rays_d[(size_t)n * 3 + 1], dz = rays_d[(size_t)n * 3 + 2];
    const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float Bh = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
    const float Cc = fmaf(oz, oz, fmaf(oy, oy, ox * ox)) - radius * radius;
    const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;
    const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
    const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
    const float phi = atan2f(z, x);
    coords[(size_t)n * 2] = 2.0f * theta * kRPi - 1.0f;
    coords[(size_t)n * 2 + 1] = phi * kRPi;
}

__global__ void morton3D_kernel(const int32_t *__restrict__ coords, uint32_t N, int32_t *__restrict__ indices)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3((uint32_t)coords[(size_t)n * 3], (uint32_t)coords[(size_t)n * 3 + 1],
                                  (uint32_t)coords[(size_t)n * 3 + 2]);
}

__global__ void morton3D_invert_kernel(const int32_t *__restrict__ indices, uint32_t N, int32_t *__restrict__ coords)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t v = (uint32_t)indices[n];
    coords[(size_t)n * 3] = (int32_t)compact_bits(v);
    coords[(size_t)n * 3 + 1] = (int32_t)compact_bits(v >> 1);
    coords[(size_t)n * 3 + 2] = (int32_t)compact_bits(v >> 2);
}

// one output byte per lane: two 16-byte loads in, one byte out (wave writes 64 contiguous bytes)
__global__ void packbits_kernel(const float *__restrict__ grid, uint32_t N, float thresh, uint8_t *__restrict__ bitfield)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
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

// sample -> ray id.  The reference fills each ray's segment from one lane; here a whole wave owns
// a ray and lane k fills elements k, k+64, ... of its segment (coalesced stores).
__global__ void flatten_rays_kernel(const int32_t *__restrict__ rays, uint32_t N, uint32_t M, int32_t *__restrict__ res)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    if (wave >= N) return;
    const uint32_t off = (uint32_t)rays[(size_t)wave * 2], cnt = (uint32_t)rays[(size_t)wave * 2 + 1];
    for (uint32_t i = lane; i < cnt; i += kWave)
        if (off + i < M) res[off + i] = (int32_t)wave;
}

// ------------------------------------------------------------------ the stepping rule
struct Marcher {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Hf, cascades, Hm1;
    bool contract;
    const uint8_t *__restrict__ grid;

    __device__ __forceinline__ void setup(const float *__restrict__ o, const float *__restrict__ d, bool inference,
                                          float bound_, bool contract_, float dt_gamma_, uint32_t max_steps,
                                          uint32_t C, uint32_t H, const uint8_t *__restrict__ grid_)
    {
        ox = o[0]; oy = o[1]; oz = o[2];
        dx = d[0]; dy = d[1]; dz = d[2];
        if (inference) {
            rdx = 1.0f / (dx + 1e-10f); rdy = 1.0f / (dy + 1e-10f); rdz = 1.0f / (dz + 1e-10f);
        } else {
            rdx = 1.0f / dx; rdy = 1.0f / dy; rdz = 1.0f / dz;
        }
        bound = bound_;
        contract = contract_;
        dt_gamma = dt_gamma_;
        dt_min = 2.0f * kSqrt3 / (float)max_steps;
        dt_max = 2.0f * kSqrt3 * bound_ / (float)H;
        Hf = (float)H;
        Hm1 = (float)(H - 1u);
        rH = 1.0f / (float)H;
        H3 = (float)(H * H * H);
        cascades = (float)C;
        grid = grid_;
    }

    // One evaluation at parameter t.  true: a sample starts at t (p = contracted position, dt its
    // length; the caller advances t).  false: empty space was skipped, t has been advanced.
    __device__ __forceinline__ bool probe(float &t, float &dt_out, float &px, float &py, float &pz) const
    {
        const uint8_t *__restrict__ g = grid;
        return probe_with(t, dt_out, px, py, pz, [g](uint32_t bit) { return (g[bit >> 3] >> (bit & 7u)) & 1u; });
    }

    template <class Occ>
    __device__ __forceinline__ bool probe_with(float &t, float &dt_out, float &px, float &py, float &pz, Occ occupied) const
    {
        float tt;
        if (classify(t, dt_out, px, py, pz, tt, occupied)) return true;
        do {
            const float dt = clampf(t * dt_gamma, dt_min, dt_max);
            t += dt;
        } while (t < tt);
        return false;
    }

    // What happens at parameter t: true = a sample starts here (p, dt as in probe); false = the cell is empty and
    // the ray leaves it at parameter tt (the caller steps t forward until t >= tt).
    template <class Occ>
    __device__ __forceinline__ bool classify(float t, float &dt_out, float &px, float &py, float &pz, float &tt,
                                             Occ occupied) const
    {
        const float x = clampf(fmaf(t, dx, ox), -bound, bound);
        const float y = clampf(fmaf(t, dy, oy), -bound, bound);
        const float z = clampf(fmaf(t, dz, oz), -bound, bound);
        const float dt = clampf(t * dt_gamma, dt_min, dt_max);

        const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        int level = 0;   // one cascade: both mip_of() clamp to 0
        if (cascades > 1.0f) level = max(mip_of(mag, cascades), mip_of(dt * Hf * 0.5f, cascades));
        const float mip_bound = fminf(scalbnf(1.0f, level), bound);
        const float mip_rbound = 1.0f / mip_bound;

        float cx = x, cy = y, cz = z;
        const bool outside = contract && mag > 1.0f;
        if (outside) {
            const float s = (2.0f - 1.0f / mag) / mag;
            cx *= s; cy *= s; cz *= s;
        }
        const int nx = (int)clampf(0.5f * fmaf(cx, mip_rbound, 1.0f) * Hf, 0.0f, Hm1);
        const int ny = (int)clampf(0.5f * fmaf(cy, mip_rbound, 1.0f) * Hf, 0.0f, Hm1);
        const int nz = (int)clampf(0.5f * fmaf(cz, mip_rbound, 1.0f) * Hf, 0.0f, Hm1);

        const uint32_t bit = (uint32_t)((float)level * H3 + (float)morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        const bool occ = occupied(bit);

        if (occ || outside) {
            px = cx; py = cy; pz = cz;
            dt_out = dt;
            return true;
        }
        const float sx = copysignf(1.0f, dx), sy = copysignf(1.0f, dy), sz = copysignf(1.0f, dz);
        const float tx = fmaf(fmaf(((float)nx + 0.5f + 0.5f * sx) * rH, 2.0f, -1.0f), mip_bound, -cx) * rdx;
        const float ty = fmaf(fmaf(((float)ny + 0.5f + 0.5f * sy) * rH, 2.0f, -1.0f), mip_bound, -cy) * rdy;
        const float tz = fmaf(fmaf(((float)nz + 0.5f + 0.5f * sz) * rH, 2.0f, -1.0f), mip_bound, -cz) * rdz;
        tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        return false;
    }

    // Position / step of the sample that STARTS at t (pure function of the ray and t); used by
    // the arena expansion, which re-derives what probe() returned when it emitted the sample.
    __device__ __forceinline__ void sample_at(float t, float &dt, float &px, float &py, float &pz) const
    {
        const float x = clampf(fmaf(t, dx, ox), -bound, bound);
        const float y = clampf(fmaf(t, dy, oy), -bound, bound);
        const float z = clampf(fmaf(t, dz, oz), -bound, bound);
        dt = clampf(t * dt_gamma, dt_min, dt_max);
        const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        px = x; py = y; pz = z;
        if (contract && mag > 1.0f) {
            const float s = (2.0f - 1.0f / mag) / mag;
            px *= s; py *= s; pz *= s;
        }
    }
};

// ------------------------------------------------------------------ training march (reference protocol)
// pass 1: count samples per ray (lane = ray).  MODE 0 writes rays[n,1]; MODE 1 additionally stores
// each sample's start time to t_scratch[n*max_steps + k] (arena variant).
template <int MODE>
__global__ __launch_bounds__(kRayBlock) void march_count_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, const uint8_t *__restrict__ grid, float bound,
    bool contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
    const float *__restrict__ nears, const float *__restrict__ fars, const float *__restrict__ noises,
    int32_t *__restrict__ rays, float *__restrict__ t_scratch)
{
    const uint32_t n = blockIdx.x * kRayBlock + threadIdx.x;
    if (n >= N) return;
    Marcher m;
    m.setup(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, false, bound, contract, dt_gamma, max_steps, C, H, grid);
    const float far = fars[n];
    float t = nears[n];
    t = fmaf(clampf(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
    uint32_t step = 0;
    float *slab = MODE == 1 ? t_scratch + (size_t)n * max_steps : nullptr;
    while (t < far && step < max_steps) {
        float dt, px, py, pz;
        if (m.probe(t, dt, px, py, pz)) {
            if (MODE == 1) slab[step] = t;
            t += dt;
            step++;
        }
    }
    rays[(size_t)n * 2 + 1] = (int32_t)step;
}

// ------------------------------------------------------------------ occupancy index (LDS-resident march)
// The march is a chain of dependent probes (~600 per ray) and runs one wave per SIMD, so its time is
// probes x (ALU + bitfield load latency); an L2 hit costs several hundred cycles, an LDS read ~100.  The 128^3
// bitfield (256 KiB) does not fit the 160 KiB LDS, but a trained grid is sparse and Morton-ordered: 64
// consecutive bits are one 4x4x4 block.  Index (uint32 words, built after every packbits):
//   [0] nnz   [1] n_words64   [2..3] reserved
//   pairs[n_words64 / 32] x {mask32: which of the group's 32 blocks are non-zero, rank: non-zero blocks before the group}
//   blocks[nnz] x uint64, in block order
// A probe is one 8-byte LDS read (pair) and, for a non-empty block, a second one (the block).  Same bits, so the
// march result is unchanged.  When nnz exceeds the LDS budget (early training: grid all occupied) the kernel
// falls back to global-memory probes.
constexpr uint32_t kOccHeader = 4;

// One workgroup (the ranks are a prefix over all groups).  A wave takes a contiguous run of groups and reads each group's
// 32 words with one coalesced load (lanes 0..31: group g, lanes 32..63: group g + 1), eight loads in flight: the first
// version had every lane walk its own group word by word -- 64 cache lines per load instruction, 17 us for 256 KiB.
__global__ __launch_bounds__(1024) void occupancy_index_kernel(const uint8_t *__restrict__ grid, uint32_t n_words64,
                                                              uint32_t *__restrict__ index)
{
    __shared__ uint32_t wave_sum[16];
    const uint64_t *__restrict__ words = reinterpret_cast<const uint64_t *>(grid);
    const uint32_t n_groups = n_words64 >> 5;
    uint32_t *pairs = index + kOccHeader;
    uint64_t *blocks = reinterpret_cast<uint64_t *>(index + kOccHeader + 2 * (size_t)n_groups);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6, half = lane >> 5, l32 = lane & 31u;
    // wave `wid` owns groups [g0, g1), two per step
    const uint32_t per = ((n_groups + 15u) / 16u + 1u) & ~1u, g0 = min(n_groups, wid * per), g1 = min(n_groups, g0 + per);
    uint32_t mine = 0;
    for (uint32_t g = g0; g < g1; g += 16u) {
        uint64_t wv[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t gg = g + 2u * k + half;
            wv[k] = gg < g1 ? words[(size_t)gg * 32 + l32] : 0ull;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const unsigned long long b = __ballot(wv[k] != 0ull);
            const uint32_t gg = g + 2u * k + half;
            const uint32_t mask = half ? (uint32_t)(b >> 32) : (uint32_t)b;
            if (l32 == 0 && gg < g1) pairs[2 * gg] = mask;
            mine += (g + 2u * k < g1 ? __popc((uint32_t)b) : 0u) + (g + 2u * k + 1u < g1 ? __popc((uint32_t)(b >> 32)) : 0u);
        }
    }
    // (`mine` is wave-uniform: the wave's number of non-empty blocks)
    if (lane == 0) wave_sum[wid] = mine;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (uint32_t k = 0; k < 16u; k++) {
        if (k < wid) run += wave_sum[k];
        total += wave_sum[k];
    }
    for (uint32_t g = g0; g < g1; g += 16u) {
        uint64_t wv[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t gg = g + 2u * k + half;
            wv[k] = gg < g1 ? words[(size_t)gg * 32 + l32] : 0ull;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const unsigned long long b = __ballot(wv[k] != 0ull);
            const uint32_t gg = g + 2u * k + half;
            const uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
            const uint32_t first = run + (half ? __popc(lo) : 0u);        // rank of this lane's group
            if (l32 == 0 && gg < g1) pairs[2 * gg + 1] = first;
            if (wv[k] != 0ull) blocks[first + __popc((half ? hi : lo) & ((1u << l32) - 1u))] = wv[k];
            run += __popc(lo) + __popc(hi);
        }
    }
    if (tid == 0) {
        index[0] = total;
        index[1] = n_words64;
        index[2] = index[3] = 0;
    }
}

constexpr uint32_t kFastBlock = 256;

// arena pass 1 with the occupancy index staged in LDS (lane = ray, 4 waves per workgroup share one copy)
__global__ __launch_bounds__(kFastBlock) void march_count_indexed_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, const uint8_t *__restrict__ grid,
    const uint32_t *__restrict__ index, uint32_t lds_blocks_cap, float bound, bool contract, float dt_gamma,
    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float *__restrict__ nears,
    const float *__restrict__ fars, const float *__restrict__ noises, int32_t *__restrict__ rays,
    float *__restrict__ t_scratch)
{
    extern __shared__ uint64_t occ_lds[];
    const uint32_t nnz = index[0], n_groups = index[1] >> 5;
    const bool staged = nnz <= lds_blocks_cap;   // uniform over the grid
    if (staged) {
        // pairs and blocks are contiguous in the index: one flat copy of n_groups + nnz 8-byte words
        const uint64_t *__restrict__ src = reinterpret_cast<const uint64_t *>(index + kOccHeader);
        const uint32_t n64 = n_groups + nnz;
        for (uint32_t i = threadIdx.x; i < n64; i += kFastBlock) occ_lds[i] = src[i];
    }
    __syncthreads();
    const uint32_t n = blockIdx.x * kFastBlock + threadIdx.x;
    if (n >= N) return;
    Marcher m;
    m.setup(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, false, bound, contract, dt_gamma, max_steps, C, H, grid);
    const float far = fars[n];
    float t = nears[n];
    t = fmaf(clampf(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
    uint32_t step = 0;
    float *slab = t_scratch + (size_t)n * max_steps;
    const uint64_t *pairs = occ_lds, *blocks = occ_lds + n_groups;
    auto in_lds = [pairs, blocks](uint32_t bit) -> uint32_t {
        const uint32_t w = bit >> 6, k = w & 31u;
        const uint64_t pr = pairs[w >> 5];
        const uint32_t mask = (uint32_t)pr;
        if (!((mask >> k) & 1u)) return 0u;
        const uint32_t idx = (uint32_t)(pr >> 32) + __popc(mask & ((1u << k) - 1u));
        return (uint32_t)(blocks[idx] >> (bit & 63u)) & 1u;
    };
    if (staged) {
        while (t < far && step < max_steps) {
            float dt, px, py, pz;
            if (m.probe_with(t, dt, px, py, pz, in_lds)) {
                slab[step] = t;
                t += dt;
                step++;
            }
        }
    } else {
        while (t < far && step < max_steps) {
            float dt, px, py, pz;
            if (m.probe(t, dt, px, py, pz)) {
                slab[step] = t;
                t += dt;
                step++;
            }
        }
    }
    rays[(size_t)n * 2 + 1] = (int32_t)step;
}

// ------------------------------------------------------------------ chain-parallel march (arena pass 1)
// The reference loop visits parameters t_0, t_1, ... with t_{k+1} = t_k + clamp(t_k * dt_gamma, dt_min, dt_max): the
// SAME recurrence whether it emits a sample or skips empty space, so a ray's chain of candidate parameters does
// not depend on the occupancy grid.  Only which elements are visited does: an occupied element is a sample and moves on
// to the next element; an empty one jumps to the first element >= the exit of its cell.  That splits the march into
//   chain     lane = ray: generate the chain (one add per element), stored ray-major (transposed through LDS)
//   classify  thread = (element, ray): cell lookup + exit parameter -> code[k] = 0 for a sample, else the jump length.
//             ~2.5 M independent probes on the whole chip instead of ~600 dependent ones on 64 waves
//   walk      wave = ray: follow the codes through 64-element windows (runs of samples are emitted by all lanes at
//             once, jumps read one lane) and write the sample start times the expansion kernel consumes
// Results are identical to the serial loop (same float expressions on the same t values).
__global__ __launch_bounds__(kRayBlock) void march_chain_kernel(const float *__restrict__ nears,
                                                               const float *__restrict__ fars,
                                                               const float *__restrict__ noises, float dt_gamma,
                                                               float dt_min, float dt_max, uint32_t N, uint32_t chain_cap,
                                                               float *__restrict__ chain, int32_t *__restrict__ chain_len,
                                                               int32_t *__restrict__ counter)
{
    // Layout: RAY-major, chain[ray * chain_cap + element] (as the codes below): the walk kernel reads 64 consecutive
    // elements of ONE ray per load -- element-major that was 64 cache lines per load instruction, 4.9 M line requests per
    // batch, as many as the encoder's forward pass issues, on the stream that runs beside the step's MLP kernels.
    // The generator itself is one lane per ray, so a block of 64 elements x 64 rays is transposed through LDS and leaves
    // as 256-byte rows.
    __shared__ float tile[64][65];
    static_assert(kRayBlock == 64, "one wave per workgroup");
    const uint32_t lane = threadIdx.x, n0 = blockIdx.x * kRayBlock, n = n0 + lane;
    const bool have = n < N;
    const float far = have ? fars[n] : 0.0f;
    float t = have ? nears[n] : 0.0f;
    if (have) t = fmaf(clampf(t * dt_gamma, dt_min, dt_max), noises[n], t);
    // (elements past a lane's end are still generated and stored -- never read: consumers stop at chain_len -- and its
    // length is the number of elements that were < far, which is the same thing because t only grows)
    uint32_t len = 0;
    const uint32_t rows = min(64u, N - n0);
    for (uint32_t k0 = 0; k0 < chain_cap; k0 += 64u) {
        const unsigned long long alive = __ballot(have && t < far);   // rays whose chain reaches into this block
        if (alive == 0ull) break;
        const uint32_t cols = min(64u, chain_cap - k0);
#pragma unroll 8
        for (uint32_t e = 0; e < 64u; e++) {
            tile[e][lane] = t;
            const bool in = e < cols && t < far;
            len += in ? 1u : 0u;
            if (e < cols) t += clampf(t * dt_gamma, dt_min, dt_max);
        }
        __syncthreads();   // (one wave per workgroup: just the ordering of its own LDS traffic)
        for (uint32_t r = 0; r < rows; r++)
            if (((alive >> r) & 1ull) && lane < cols) chain[(size_t)(n0 + r) * chain_cap + k0 + lane] = tile[lane][r];
        __syncthreads();
    }
    if (have) {
        chain_len[n] = (int32_t)len;
        if (t < far) atomicOr(counter + 2, 1);   // chain buffer too short: the ray was cut (reported, never silent)
    }
}

// thread = (element, ray) with the ELEMENT index fastest: a workgroup works on 256 consecutive elements of one ray (its
// origin / direction are uniform loads; chain reads, the jump search and the code stores are contiguous)
__global__ __launch_bounds__(256) void march_classify_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, const uint8_t *__restrict__ grid, float bound,
    bool contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t chain_cap,
    const float *__restrict__ chain, const int32_t *__restrict__ chain_len, uint16_t *__restrict__ code)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t n = blockIdx.y; n < N; n += gridDim.y) {   // (more rays than grid rows: rare)
        const uint32_t len = (uint32_t)chain_len[n];
        if (k >= len) continue;
        Marcher m;
        m.setup(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, false, bound, contract, dt_gamma, max_steps, C, H, grid);
        const float *__restrict__ row = chain + (size_t)n * chain_cap;
        const float t = row[k];
        float dt, px, py, pz, tt;
        const uint8_t *__restrict__ g = grid;
        uint32_t c = 0;
        if (!m.classify(t, dt, px, py, pz, tt, [g](uint32_t bit) { return (g[bit >> 3] >> (bit & 7u)) & 1u; })) {
            uint32_t j = k + 1;
            while (j < len && row[j] < tt) j++;
            c = j - k;
        }
        code[(size_t)n * chain_cap + k] = (uint16_t)c;
    }
}

__global__ __launch_bounds__(256) void march_walk_kernel(const float *__restrict__ chain,
                                                        const int32_t *__restrict__ chain_len,
                                                        const uint16_t *__restrict__ code, uint32_t N, uint32_t chain_cap,
                                                        uint32_t max_steps, int32_t *__restrict__ rays,
                                                        float *__restrict__ t_scratch)
{
    const uint32_t n = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (n >= N) return;
    chain += (size_t)n * chain_cap;   // ray-major: this ray's elements and codes are contiguous
    code += (size_t)n * chain_cap;
    const uint32_t len = (uint32_t)chain_len[n];
    float *slab = t_scratch + (size_t)n * max_steps;
    uint32_t k = 0, step = 0;   // wave-uniform
    while (k < len && step < max_steps) {
        const uint32_t e = k + lane;
        const bool have = e < len;
        const uint32_t c = have ? (uint32_t)code[e] : 0xffffu;
        const float t = have ? chain[e] : 0.0f;
        const unsigned long long samples = __ballot(have && c == 0u);
        uint32_t pos = 0;
        while (pos < 64u && k + pos < len && step < max_steps) {
            if ((samples >> pos) & 1ull) {
                const unsigned long long rest = ~(samples >> pos);
                uint32_t run = rest ? (uint32_t)__ffsll((long long)rest) - 1u : 64u - pos;
                run = min(run, max_steps - step);
                if (lane >= pos && lane < pos + run) slab[step + (lane - pos)] = t;
                step += run;
                pos += run;
            } else {
                pos += (uint32_t)__builtin_amdgcn_readlane((int)c, (int)pos);
            }
        }
        k += pos;
    }
    if (lane == 0) rays[(size_t)n * 2 + 1] = (int32_t)step;
}

// ------------------------------------------------------------------ constant-step march (arena pass 1, dt_gamma == 0)
// With dt_gamma == 0 the recurrence is t_{k+1} = fl(t_k + d) for ONE float d, and repeated float addition of a constant has
// a closed form per binade: inside [2^e, 2^(e+1)) every t is a multiple of u = 2^(e-23); with d = (q + f) u the rounded sum
// is t + q u or t + (q + 1) u, decided by f alone -- except for f == 1/2 (a tie, broken by the parity of t / u), where the
// FIRST step of a binade may differ and every later one is the even one of q, q + 1.  So once two consecutive steps inside
// a binade are equal, every following element of that binade is   bits(t_j) = bits(t_0) + j * s   (bit patterns of positive
// floats are linear in the mantissa), up to and including the last one below 2^(e+1): the rounded sum stays in the binade
// exactly when the integer model says so (x = a + q + f < 2^24 - 1/2  <=>  a + s < 2^24 in each of the cases above).  A ray's
// chain is therefore a short table of segments {first element, its bits, step in ulps}: one run per binade plus the
// elements produced by real float additions at its edges (the first, irregular, step; the addition that crosses into the
// next binade).  ~2-3 segments per binade between d's and far's: 20-30, built by ~100 scalar-ish instructions.
// With the table, element k of the chain is two integer operations, and "the first element >= tt" (the jump out of an empty
// cell) is one division -- so chain, classify and walk become ONE kernel, wave = ray, that touches no chain or code buffer:
// the window of 64 candidates is computed, classified and walked in registers.  Same float values as the serial loop, hence
// the same samples, bit for bit (tests/test_gpu_parity.py: the chain mode with dt_gamma == 0, and the binade / tie sweep).
constexpr uint32_t kSegCap = 64;   // segments per ray; a chain that needs more is reported like a short chain buffer
constexpr uint32_t kCsRays = 16;   // rays (= waves) per workgroup: they share one LDS copy of the occupancy index

// the windows of one ray: 64 candidates at a time are computed from the table, classified and walked (the walk is scalar
// code: k, step, pos live in SGPRs).  Returns the ray's sample count; sample start times go to `slab`.
template <class Occ>
__device__ __forceinline__ uint32_t const_step_windows(const Marcher &m, Occ occupied, const uint32_t *seg_k,
                                                       const uint32_t *seg_b, const uint32_t *seg_s, uint32_t ns,
                                                       uint32_t len, uint32_t max_steps, float *__restrict__ slab,
                                                       uint32_t lane)
{
    uint32_t k = 0, step = 0, si = 0;   // wave-uniform
    while (k < len && step < max_steps) {
        while ((uint32_t)__builtin_amdgcn_readfirstlane((int)seg_k[si + 1]) <= k) si++;
        const uint32_t e = k + lane;
        const bool have = e < len;
        uint32_t c = 0xffffffffu;
        float tk = 0.0f;
        if (have) {
            uint32_t i = si;
            while (seg_k[i + 1] <= e) i++;
            tk = __uint_as_float(seg_b[i] + (e - seg_k[i]) * seg_s[i]);
            float dt, px, py, pz, tt;
            if (m.classify(tk, dt, px, py, pz, tt, occupied)) {
                c = 0u;
            } else {
                // the serial loop steps until t >= tt: the first element after e that is not below tt (NaN compares false:
                // one step), or the end of the chain
                uint32_t j = e + 1u;
                if (tt == tt) {
                    const uint32_t btt = __float_as_uint(tt);   // tt >= t >= 0: positive floats order like their bits
                    j = len;
                    for (uint32_t ii = i; ii < ns; ii++) {
                        const uint32_t k0 = seg_k[ii], k1 = seg_k[ii + 1], b0 = seg_b[ii], s = seg_s[ii];
                        uint32_t jj = k0;
                        if (btt > b0) jj = s ? k0 + min((btt - b0 + s - 1u) / s, k1 - k0) : k1;
                        jj = max(jj, e + 1u);
                        if (jj < k1) {
                            j = jj;
                            break;
                        }
                    }
                }
                c = j - e;
            }
        }
        // The walk.  Serially: pos = 0; while pos is inside the window: a sample is emitted and pos += 1, an empty candidate
        // jumps, pos += c.  Followed hop by hop that is ~15 dependent scalar iterations per window, and with sixteen rays per
        // CU the ONE scalar unit of the CU was the kernel's limit (3 200 SALU against 1 600 VALU instructions per ray).  Here
        // the visited positions are computed by pointer doubling instead: with T(i) = the position after lane i (64 = left
        // the window) and P_r = T applied 2^r times, lane m composes the P_r of its binary digits and ends with the m-th
        // visited position o_m -- six rounds of two cross-lane gathers, no scalar loop.
        uint32_t T = 64u;
        if (have) T = min(lane + (c ? min(c, 64u) : 1u), 64u);
        uint32_t P = T, o = 0;
#pragma unroll
        for (uint32_t r = 0; r < 6u; r++) {
            const uint32_t Po = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(min(o, 63u) << 2), (int)P);
            if ((lane >> r) & 1u) o = o < 64u ? Po : 64u;
            if (r < 5u) {
                const uint32_t PP = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(min(P, 63u) << 2), (int)P);
                P = P < 64u ? PP : 64u;
            }
        }
        // o_0 < o_1 < ... while inside the window (and inside the chain: a jump to the chain's end lands on len - k)
        const uint32_t lim = min(64u, len - k);
        const bool visited = o < lim;
        const uint32_t oc = min(o, 63u) << 2;
        const float t_o = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)oc, (int)__float_as_uint(tk)));
        const uint32_t c_o = (uint32_t)__builtin_amdgcn_ds_bpermute((int)oc, (int)c);
        const unsigned long long vis = __ballot(visited), smp = __ballot(visited && c_o == 0u);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(smp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)smp, 0u));
        const uint32_t room = max_steps - step, total = (uint32_t)__popcll(smp);
        if (visited && c_o == 0u && rank < room) slab[step + rank] = t_o;   // samples in order: one coalesced store
        step += min(total, room);
        // where the last visited position leads (not capped: it is the next window's first candidate)
        const uint32_t last = (uint32_t)__popcll(vis) - 1u;   // (lane 0 is always visited: o_0 = 0 < lim)
        k += (uint32_t)__builtin_amdgcn_readlane((int)(o + (c_o ? c_o : 1u)), (int)last);
    }
    return step;
}

// index: the occupancy index of occupancy_index_kernel or NULL.  When its non-empty blocks fit `lds_blocks_cap` the
// workgroup stages it in LDS and every probe is one or two LDS reads instead of a byte load from L2 -- the kernel runs
// beside the encoder's forward pass, which is bound by exactly those L2 requests; otherwise (early training: every cell
// occupied) the probes go to the bitfield in global memory.  Same bits either way.
__global__ __launch_bounds__(kCsRays * 64) void march_const_step_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, const uint8_t *__restrict__ grid,
    const uint32_t *__restrict__ index, uint32_t lds_blocks_cap, float bound, bool contract, uint32_t max_steps, uint32_t N,
    uint32_t C, uint32_t H, const float *__restrict__ nears, const float *__restrict__ fars,
    const float *__restrict__ noises, uint32_t chain_cap, int32_t *__restrict__ rays, float *__restrict__ t_scratch,
    int32_t *__restrict__ counter)
{
    extern __shared__ uint64_t occ_lds[];
    __shared__ uint32_t seg_k[kCsRays][kSegCap + 1], seg_b[kCsRays][kSegCap], seg_s[kCsRays][kSegCap];
    // (readfirstlane: the wave index, and with it everything that depends on the ray alone, in scalar registers)
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint32_t n = blockIdx.x * kCsRays + w;
    const bool ray = n < N;
    uint32_t n_groups = 0;
    bool staged = false;
    if (index) {
        const uint32_t nnz = index[0];
        n_groups = index[1] >> 5;
        staged = nnz <= lds_blocks_cap;   // uniform over the grid
        if (staged) {   // pairs and blocks are contiguous in the index: one flat copy of n_groups + nnz 8-byte words
            const uint64_t *__restrict__ src = reinterpret_cast<const uint64_t *>(index + kOccHeader);
            for (uint32_t i = threadIdx.x; i < n_groups + nnz; i += kCsRays * 64u) occ_lds[i] = src[i];
        }
    }
    Marcher m;
    m.setup(rays_o + (size_t)(ray ? n : 0u) * 3, rays_d + (size_t)(ray ? n : 0u) * 3, false, bound, contract, 0.0f, max_steps, C,
            H, grid);
    // ---- the segment table (every lane computes the same values; lane 0 stores them)
    const float d = clampf(0.0f, m.dt_min, m.dt_max);
    const float far = ray ? fars[n] : 0.0f;
    float t = ray ? nears[n] : 0.0f;
    if (ray) t = fmaf(d, noises[n], t);
    uint32_t len = 0, ns = 0;
    bool cut = false;
    // (a loop every lane runs with the same values: the ballot says so to the compiler, which then keeps its counters scalar)
    while (__ballot(t < far && len < chain_cap) != 0ull) {
        const uint32_t bt = __float_as_uint(t);
        if (ns == kSegCap || (bt >> 31)) {   // (negative parameters: not a case the callers produce -- near >= min_near >= 0)
            cut = true;
            break;
        }
        const float t1 = t + d;
        const uint32_t b1 = __float_as_uint(t1);
        uint32_t s = 0, cnt = 1;
        float tn = t1;
        if ((bt >> 23) == (b1 >> 23) && b1 > bt) {
            const uint32_t b2 = __float_as_uint(t1 + d);
            if ((b2 >> 23) == (bt >> 23) && b2 - b1 == b1 - bt) {   // two equal steps inside the binade: the steady state
                s = b1 - bt;
                const uint32_t top = ((bt >> 23) + 1u) << 23, bf = __float_as_uint(far);   // far > t >= 0
                cnt = (top - 1u - bt) / s + 1u;
                if (bf < top) cnt = min(cnt, (bf - bt + s - 1u) / s);   // elements below far
                cnt = min(cnt, chain_cap - len);
                tn = __uint_as_float(bt + (cnt - 1u) * s) + d;
            }
        }
        if (lane == 0) {
            seg_k[w][ns] = len;
            seg_b[w][ns] = bt;
            seg_s[w][ns] = s;
        }
        ns++;
        len += cnt;
        t = tn;
    }
    if (lane == 0) {
        seg_k[w][ns] = len;
        if (ray && (cut || t < far)) atomicOr(counter + 2, 1);   // cut short: reported, never silent (as march_chain_kernel)
    }
    __syncthreads();
    if (!ray) return;
    len = (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
    ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)ns);
    float *slab = t_scratch + (size_t)n * max_steps;
    uint32_t step;
    if (staged) {
        const uint64_t *pairs = occ_lds, *blocks = occ_lds + n_groups;
        step = const_step_windows(m, [pairs, blocks](uint32_t bit) -> uint32_t {
            const uint32_t wd = bit >> 6, k = wd & 31u;
            const uint64_t pr = pairs[wd >> 5];
            const uint32_t mask = (uint32_t)pr;
            if (!((mask >> k) & 1u)) return 0u;
            const uint32_t idx = (uint32_t)(pr >> 32) + __popc(mask & ((1u << k) - 1u));
            return (uint32_t)(blocks[idx] >> (bit & 63u)) & 1u;
        }, seg_k[w], seg_b[w], seg_s[w], ns, len, max_steps, slab, lane);
    } else {
        const uint8_t *__restrict__ g = grid;
        step = const_step_windows(m, [g](uint32_t bit) -> uint32_t { return (g[bit >> 3] >> (bit & 7u)) & 1u; }, seg_k[w],
                                  seg_b[w], seg_s[w], ns, len, max_steps, slab, lane);
    }
    if (lane == 0) rays[(size_t)n * 2 + 1] = (int32_t)step;
}

// Exclusive prefix sum of rays[:,1] into rays[:,0] in ray order; one workgroup of 1024 lanes
// walks the rays in tiles (wave shuffle scan + one LDS hop per tile).
//   reference protocol (arena == false): offsets start at the incoming counter[0]; counter[0] = end.
//   arena (M_cap > 0): offsets start at 0; a ray is kept while offset + count <= M_cap (offsets are
//   monotone, so the kept rays are a prefix of the batch); dropped rays get count 0.
//   counter[0] = samples of the kept rays (<= M_cap), counter[1] = samples the batch needed.
__global__ __launch_bounds__(1024) void march_scan_kernel(int32_t *__restrict__ rays, uint32_t N,
                                                         int32_t *__restrict__ counter, uint32_t M_cap,
                                                         bool arena)
{
    __shared__ uint32_t wave_sum[16], wave_kept[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (tid == 0) carry_s = arena ? 0u : (uint32_t)counter[0];
    __syncthreads();
    uint32_t kept_total = 0;   // meaningful in lane 0 of wave 0
    // 4 consecutive rays per lane: a 4096-ray batch is one tile (one load round trip, one store round trip)
    for (uint32_t base = 0; base < N; base += 4096u) {
        const uint32_t n0 = base + tid * 4u;
        uint32_t cnt[4], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            cnt[j] = n0 + j < N ? (uint32_t)rays[(size_t)(n0 + j) * 2 + 1] : 0u;
            sum += cnt[j];
        }
        uint32_t v = sum;  // inclusive scan inside the wave
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t up = __shfl_up(v, d, 64);
            if (lane >= d) v += up;
        }
        if (lane == 63u) wave_sum[wid] = v;
        __syncthreads();
        uint32_t wave_off = 0, tile_total = 0;
        for (uint32_t w = 0; w < 16u; w++) {
            if (w < wid) wave_off += wave_sum[w];
            tile_total += wave_sum[w];
        }
        const uint32_t carry = carry_s;
        uint32_t off = carry + wave_off + v - sum, kept = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t n = n0 + j;
            if (n < N) {
                if (!arena) {
                    rays[(size_t)n * 2] = (int32_t)off;
                } else {
                    const bool keep = off + cnt[j] <= M_cap;
                    rays[(size_t)n * 2] = (int32_t)(keep ? off : M_cap);
                    if (keep)
                        kept += cnt[j];
                    else
                        rays[(size_t)n * 2 + 1] = 0;
                }
            }
            off += cnt[j];
        }
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1) kept += __shfl_xor(kept, d, 64);
        if (lane == 0) wave_kept[wid] = kept;
        __syncthreads();
        if (tid == 0) {
            carry_s = carry + tile_total;
            for (uint32_t w = 0; w < 16u; w++) kept_total += wave_kept[w];
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (arena) {
            counter[0] = (int32_t)kept_total;
            counter[1] = (int32_t)carry_s;
        } else {
            counter[0] = (int32_t)carry_s;
        }
    }
}

// pass 2 of the reference protocol: re-march and write (lane = ray, like the reference).
__global__ __launch_bounds__(kRayBlock) void march_write_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ rays_ldir,
    const uint8_t *__restrict__ grid, float bound, bool contract, float dt_gamma, uint32_t max_steps, uint32_t N,
    uint32_t C, uint32_t H, const float *__restrict__ nears, const float *__restrict__ fars,
    const float *__restrict__ noises, const int32_t *__restrict__ rays, float *__restrict__ xyzs,
    float *__restrict__ dirs, float *__restrict__ ts, float *__restrict__ ldirs)
{
    const uint32_t n = blockIdx.x * kRayBlock + threadIdx.x;
    if (n >= N) return;
    Marcher m;
    m.setup(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, false, bound, contract, dt_gamma, max_steps, C, H, grid);
    const uint32_t off = (uint32_t)rays[(size_t)n * 2], budget = (uint32_t)rays[(size_t)n * 2 + 1];
    float lx = 0, ly = 0, lz = 0;
    if (rays_ldir) {
        lx = rays_ldir[(size_t)n * 3]; ly = rays_ldir[(size_t)n * 3 + 1]; lz = rays_ldir[(size_t)n * 3 + 2];
    }
    const float far = fars[n];
    float t = nears[n];
    t = fmaf(clampf(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
    uint32_t step = 0;
    while (t < far && step < budget) {
        float dt, px, py, pz;
        if (m.probe(t, dt, px, py, pz)) {
            t += dt;
            const size_t i = (size_t)off + step;
            xyzs[i * 3] = px; xyzs[i * 3 + 1] = py; xyzs[i * 3 + 2] = pz;
            dirs[i * 3] = m.dx; dirs[i * 3 + 1] = m.dy; dirs[i * 3 + 2] = m.dz;
            ts[i * 2] = t; ts[i * 2 + 1] = dt;
            if (ldirs) {
                ldirs[i * 3] = lx; ldirs[i * 3 + 1] = ly; ldirs[i * 3 + 2] = lz;
            }
            step++;
        }
    }
}

// arena expansion: one WAVE per ray, lanes stride over that ray's samples; every output array is
// written with consecutive lanes on consecutive samples (coalesced), nothing is re-marched.
__global__ __launch_bounds__(256) void march_expand_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ rays_ldir,
    float bound, bool contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
    const int32_t *__restrict__ rays, const float *__restrict__ t_scratch, float *__restrict__ xyzs,
    float *__restrict__ dirs, float *__restrict__ ts, float *__restrict__ ldirs, int32_t *__restrict__ ray_idx)
{
    const uint32_t n = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[(size_t)n * 2], cnt = (uint32_t)rays[(size_t)n * 2 + 1];
    if (cnt == 0) return;
    Marcher m;
    m.setup(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, false, bound, contract, dt_gamma, max_steps, C, H, nullptr);
    float lx = 0, ly = 0, lz = 0;
    if (rays_ldir) {
        lx = rays_ldir[(size_t)n * 3]; ly = rays_ldir[(size_t)n * 3 + 1]; lz = rays_ldir[(size_t)n * 3 + 2];
    }
    const float *slab = t_scratch + (size_t)n * max_steps;
    for (uint32_t k = lane; k < cnt; k += kWave) {
        const float t = slab[k];
        float dt, px, py, pz;
        m.sample_at(t, dt, px, py, pz);
        const size_t i = (size_t)off + k;
        xyzs[i * 3] = px; xyzs[i * 3 + 1] = py; xyzs[i * 3 + 2] = pz;
        dirs[i * 3] = m.dx; dirs[i * 3 + 1] = m.dy; dirs[i * 3 + 2] = m.dz;
        ts[i * 2] = t + dt; ts[i * 2 + 1] = dt;
        if (ldirs) {
            ldirs[i * 3] = lx; ldirs[i * 3 + 1] = ly; ldirs[i * 3 + 2] = lz;
        }
        if (ray_idx) ray_idx[i] = (int32_t)n;
    }
}

// ------------------------------------------------------------------ compositing (training)
__global__ __launch_bounds__(kRayBlock) void composite_train_forward_kernel(
    const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ ts,
    const int32_t *__restrict__ rays, uint32_t M, uint32_t N, float T_thresh, float *__restrict__ weights,
    float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image)
{
    const uint32_t n = blockIdx.x * kRayBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[(size_t)n * 2], cnt = (uint32_t)rays[(size_t)n * 2 + 1];
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
    if (cnt != 0 && off + cnt <= M) {
        for (uint32_t i = off; i < off + cnt; i++) {
            const float2 tt = reinterpret_cast<const float2 *>(ts)[i];
            const float alpha = 1.0f - __expf(-sigmas[i] * tt.y);
            const float w = alpha * T;
            weights[i] = w;
            r = fmaf(w, rgbs[(size_t)i * 3], r);
            g = fmaf(w, rgbs[(size_t)i * 3 + 1], g);
            b = fmaf(w, rgbs[(size_t)i * 3 + 2], b);
            ws += w;
            d = fmaf(w, tt.x, d);
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
        }
    }
    weights_sum[n] = ws;
    depth[n] = d;
    image[(size_t)n * 3] = r;
    image[(size_t)n * 3 + 1] = g;
    image[(size_t)n * 3 + 2] = b;
}

__global__ __launch_bounds__(kRayBlock) void composite_train_backward_kernel(
    const float *__restrict__ grad_weights, const float *__restrict__ grad_weights_sum,
    const float *__restrict__ grad_depth, const float *__restrict__ grad_image, const float *__restrict__ sigmas,
    const float *__restrict__ rgbs, const float *__restrict__ ts, const int32_t *__restrict__ rays,
    const float *__restrict__ weights_sum, const float *__restrict__ depth, const float *__restrict__ image, uint32_t M,
    uint32_t N, float T_thresh, float *__restrict__ grad_sigmas, float *__restrict__ grad_rgbs)
{
    const uint32_t n = blockIdx.x * kRayBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[(size_t)n * 2], cnt = (uint32_t)rays[(size_t)n * 2 + 1];
    if (cnt == 0 || off + cnt > M) return;
    const float gr = grad_image[(size_t)n * 3], gg = grad_image[(size_t)n * 3 + 1], gb = grad_image[(size_t)n * 3 + 2];
    const float gws = grad_weights_sum[n], gd = grad_depth[n];
    const float rF = image[(size_t)n * 3], gF = image[(size_t)n * 3 + 1], bF = image[(size_t)n * 3 + 2];
    const float wsF = weights_sum[n], dF = depth[n];
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
    for (uint32_t i = off; i < off + cnt; i++) {
        const float2 tt = reinterpret_cast<const float2 *>(ts)[i];
        const float dt = tt.y, tm = tt.x;
        const float c0 = rgbs[(size_t)i * 3], c1 = rgbs[(size_t)i * 3 + 1], c2 = rgbs[(size_t)i * 3 + 2];
        const float alpha = 1.0f - __expf(-sigmas[i] * dt);
        const float w = alpha * T;
        r = fmaf(w, c0, r);
        g = fmaf(w, c1, g);
        b = fmaf(w, c2, b);
        ws += w;
        d = fmaf(w, tm, d);
        T *= 1.0f - alpha;  // the closed form below uses T after the update (raymarching.cu:681-696)
        grad_rgbs[(size_t)i * 3] = gr * w;
        grad_rgbs[(size_t)i * 3 + 1] = gg * w;
        grad_rgbs[(size_t)i * 3 + 2] = gb * w;
        float s = gr * fmaf(T, c0, -(rF - r));
        s = fmaf(gg, fmaf(T, c1, -(gF - g)), s);
        s = fmaf(gb, fmaf(T, c2, -(bF - b)), s);
        s = fmaf(gws + grad_weights[i], T - (wsF - ws), s);
        s = fmaf(gd, fmaf(T, tm, -(dF - d)), s);
        grad_sigmas[i] = dt * s;
        if (T < T_thresh) break;
    }
}

// ------------------------------------------------------------------ inference pair
__global__ __launch_bounds__(kRayBlock) void march_rays_kernel(
    uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive, const float *__restrict__ rays_t,
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, float bound, bool contract, float dt_gamma,
    uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *__restrict__ grid, const float *__restrict__ fars,
    float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ ts, const float *__restrict__ noises)
{
    const uint32_t n = blockIdx.x * kRayBlock + threadIdx.x;
    if (n >= n_alive) return;
    const uint32_t ray = (uint32_t)rays_alive[n];
    Marcher m;
    m.setup(rays_o + (size_t)ray * 3, rays_d + (size_t)ray * 3, true, bound, contract, dt_gamma, max_steps, C, H, grid);
    const float far = fars[ray];
    float t = rays_t[ray];
    t = fmaf(clampf(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
    uint32_t step = 0;
    const size_t base = (size_t)n * n_step;
    while (t < far && step < n_step) {
        float dt, px, py, pz;
        if (m.probe(t, dt, px, py, pz)) {
            t += dt;
            const size_t i = base + step;
            xyzs[i * 3] = px; xyzs[i * 3 + 1] = py; xyzs[i * 3 + 2] = pz;
            dirs[i * 3] = m.dx; dirs[i * 3 + 1] = m.dy; dirs[i * 3 + 2] = m.dz;
            ts[i * 2] = t; ts[i * 2 + 1] = dt;
            step++;
        }
    }
}

__global__ __launch_bounds__(kRayBlock) void composite_rays_kernel(
    uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *__restrict__ rays_alive, float *__restrict__ rays_t,
    const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ ts,
    float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image)
{
    const uint32_t n = blockIdx.x * kRayBlock + threadIdx.x;
    if (n >= n_alive) return;
    const uint32_t ray = (uint32_t)rays_alive[n];
    const size_t base = (size_t)n * n_step;
    float t = 0.0f;
    float d = depth[ray], r = image[(size_t)ray * 3], g = image[(size_t)ray * 3 + 1], b = image[(size_t)ray * 3 + 2];
    float ws = weights_sum[ray];
    uint32_t step = 0;
    while (step < n_step) {
        const size_t i = base + step;
        const float2 tt = reinterpret_cast<const float2 *>(ts)[i];
        if (tt.x == 0.0f) break;
        const float alpha = 1.0f - __expf(-sigmas[i] * tt.y);
        const float T = 1.0f - ws;
        const float w = alpha * T;
        ws += w;
        t = tt.x;
        d = fmaf(w, t, d);
        r = fmaf(w, rgbs[i * 3], r);
        g = fmaf(w, rgbs[i * 3 + 1], g);
        b = fmaf(w, rgbs[i * 3 + 2], b);
        if (T < T_thresh) break;
        step++;
    }
    if (step < n_step)
        rays_alive[n] = -1;
    else
        rays_t[ray] = t;
    weights_sum[ray] = ws;
    depth[ray] = d;
    image[(size_t)ray * 3] = r;
    image[(size_t)ray * 3 + 1] = g;
    image[(size_t)ray * 3 + 2] = b;
}

// ------------------------------------------------------------------ ray gradients (segment sums)
// one wave per ray: lanes stride over the ray's samples, then a butterfly reduction
__global__ __launch_bounds__(256) void march_train_backward_kernel(
    const float *__restrict__ grad_xyzs, const float *__restrict__ grad_dirs, const float *__restrict__ ts,
    const int32_t *__restrict__ rays, uint32_t N, uint32_t M, float *__restrict__ grad_rays_o,
    float *__restrict__ grad_rays_d)
{
    const uint32_t n = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[(size_t)n * 2], cnt = (uint32_t)rays[(size_t)n * 2 + 1];
    float so[3] = {0, 0, 0}, sd[3] = {0, 0, 0};
    if (off + cnt <= M) {
        for (uint32_t k = lane; k < cnt; k += kWave) {
            const size_t i = (size_t)off + k;
            const float t = ts[i * 2];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float gx = grad_xyzs[i * 3 + c];
                so[c] += gx;
                sd[c] += gx * t + (grad_dirs ? grad_dirs[i * 3 + c] : 0.0f);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1) {
            so[c] += __shfl_xor(so[c], d, 64);
            sd[c] += __shfl_xor(sd[c], d, 64);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            grad_rays_o[(size_t)n * 3 + c] = so[c];
            grad_rays_d[(size_t)n * 3 + c] = sd[c];
        }
    }
}

}  // namespace ngp

using namespace ngp;

#define NGP_1D(n, blk) dim3(ceil_div((n), (blk))), dim3(blk), 0, as_stream(stream)

extern "C" int ngp_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                                      float min_near, float *nears, float *fars, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && aabb && nears && fars, "near_far_from_aabb: null tensor");
    near_far_kernel<<<NGP_1D(N, 256u)>>>(rays_o, rays_d, aabb, N, min_near, nears, fars);
    NGP_CHECK_LAUNCH("near_far_from_aabb");
    return NGP_OK;
}

extern "C" int ngp_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords,
                                ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && coords, "sph_from_ray: null tensor");
    sph_from_ray_kernel<<<NGP_1D(N, 256u)>>>(rays_o, rays_d, radius, N, coords);
    NGP_CHECK_LAUNCH("sph_from_ray");
    return NGP_OK;
}

extern "C" int ngp_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(coords && indices, "morton3D: null tensor");
    morton3D_kernel<<<NGP_1D(N, 256u)>>>(coords, N, indices);
    NGP_CHECK_LAUNCH("morton3D");
    return NGP_OK;
}

extern "C" int ngp_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(coords && indices, "morton3D_invert: null tensor");
    morton3D_invert_kernel<<<NGP_1D(N, 256u)>>>(indices, N, coords);
    NGP_CHECK_LAUNCH("morton3D_invert");
    return NGP_OK;
}

extern "C" int ngp_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield,
                            ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(grid && bitfield, "packbits: null tensor");
    NGP_REQUIRE(((uintptr_t)grid & 15u) == 0, "packbits: grid must be 16-byte aligned");
    packbits_kernel<<<NGP_1D(N, 256u)>>>(grid, N, density_thresh, bitfield);
    NGP_CHECK_LAUNCH("packbits");
    return NGP_OK;
}

extern "C" int ngp_flatten_rays(const int32_t *rays, uint32_t N, uint32_t M, int32_t *res, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays && res, "flatten_rays: null tensor");
    flatten_rays_kernel<<<dim3(ceil_div(N, kFlatBlock / kWave)), dim3(kFlatBlock), 0, as_stream(stream)>>>(rays, N, M, res);
    NGP_CHECK_LAUNCH("flatten_rays");
    return NGP_OK;
}

extern "C" int ngp_march_rays_train(const float *rays_o, const float *rays_d, const float *rays_ldir,
                                    const uint8_t *grid, float bound, int contract, float dt_gamma,
                                    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float *nears,
                                    const float *fars, float *xyzs, float *dirs, float *ts, float *ldirs,
                                    int32_t *rays, int32_t *counter, const float *noises, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && grid && nears && fars && rays && counter && noises, "march_rays_train: null tensor");
    NGP_REQUIRE(max_steps > 0 && H > 0 && C > 0, "march_rays_train: max_steps, C and H must be positive");
    if (xyzs == nullptr) {
        march_count_kernel<0><<<NGP_1D(N, kRayBlock)>>>(rays_o, rays_d, grid, bound, contract != 0, dt_gamma, max_steps, N,
                                                       C, H, nears, fars, noises, rays, nullptr);
        march_scan_kernel<<<dim3(1), dim3(1024), 0, as_stream(stream)>>>(rays, N, counter, 0u, false);
    } else {
        NGP_REQUIRE(dirs && ts, "march_rays_train: dirs / ts missing in the write pass");
        NGP_REQUIRE(!rays_ldir || ldirs, "march_rays_train: rays_ldir given without an ldirs output");
        march_write_kernel<<<NGP_1D(N, kRayBlock)>>>(rays_o, rays_d, rays_ldir, grid, bound, contract != 0, dt_gamma,
                                                    max_steps, N, C, H, nears, fars, noises, rays, xyzs, dirs, ts,
                                                    rays_ldir ? ldirs : nullptr);
    }
    NGP_CHECK_LAUNCH("march_rays_train");
    return NGP_OK;
}

// stage 0: the whole march.  Chain-parallel variant only: 1 = its first kernel alone (the candidate parameters of every
// ray: reads rays, near/far and noise, not the occupancy grid), 2 = everything after it -- so that a caller whose grid is
// still being rebuilt can get the grid-independent third of the march out of the way
extern "C" int ngp_x_march_rays_train_arena_stage(const float *rays_o, const float *rays_d, const float *rays_ldir,
                                                  const uint8_t *grid, float bound, int contract, float dt_gamma,
                                                  uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                  const float *nears, const float *fars, const float *noises,
                                                  float *t_scratch, uint32_t M_cap, float *xyzs, float *dirs, float *ts,
                                                  float *ldirs, int32_t *rays, int32_t *counter, int32_t *ray_idx,
                                                  const uint32_t *occ_index, float *chain, uint16_t *chain_code,
                                                  int32_t *chain_len, uint32_t chain_cap, int stage, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(stage >= 0 && stage <= 2 && (stage == 0 || chain), "march_rays_train_arena: stages 1 / 2 need the chain buffers");
    NGP_REQUIRE(rays_o && rays_d && grid && nears && fars && rays && counter && noises && t_scratch && xyzs && dirs && ts,
                "march_rays_train_arena: null tensor");
    NGP_REQUIRE(max_steps > 0 && H > 0 && C > 0 && M_cap > 0, "march_rays_train_arena: bad sizes");
    const uint64_t n_bits = (uint64_t)C * H * H * H;
    if (chain) {
        NGP_REQUIRE(chain_code && chain_len && chain_cap > 0 && chain_cap < 65536u,
                    "march_rays_train_arena: chain buffers incomplete (chain_cap must be in 1..65535)");
        const float dt_min = 2.0f * kSqrt3 / (float)max_steps, dt_max = 2.0f * kSqrt3 * bound / (float)H;
        // dt_gamma == 0: the chain has a closed form and the three kernels are one (march_const_step_kernel); stage 1 then
        // has nothing to do.  NGP_MARCH_CONST_STEP=0 keeps the general kernels (A/B runs, tests of both)
        static const bool const_step_on = !(getenv("NGP_MARCH_CONST_STEP") && getenv("NGP_MARCH_CONST_STEP")[0] == '0');
        const bool const_step = const_step_on && dt_gamma == 0.0f;
        if (stage != 2 && !const_step)
            march_chain_kernel<<<NGP_1D(N, kRayBlock)>>>(nears, fars, noises, dt_gamma, dt_min, dt_max, N, chain_cap, chain,
                                                        chain_len, counter);
        if (stage == 1) {
            NGP_CHECK_LAUNCH("march_rays_train_arena");
            return NGP_OK;
        }
        if (const_step) {
            // with an occupancy index: staged in LDS when it fits.  80 KiB (+ 12 of tables), not all 160: it runs beside the main
            // stream's kernels and must leave room for their workgroups on its CUs
            constexpr uint32_t kLdsBudget = 80u * 1024u;
            uint32_t lds = 0, cap = 0;
            if (occ_index) {
                NGP_REQUIRE(n_bits % 2048u == 0 && ((uintptr_t)occ_index & 7u) == 0,
                            "march_rays_train_arena: occupancy index needs C*H^3 to be a multiple of 2048 and an 8-byte aligned buffer");
                const uint32_t n_groups = (uint32_t)(n_bits / 2048u);
                if ((size_t)n_groups * 8 + 8 * 1024 <= kLdsBudget) {
                    lds = kLdsBudget;
                    cap = (kLdsBudget - n_groups * 8u) / 8u;
                    static bool attr_set = false;
                    if (!attr_set) {
                        NGP_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void *>(march_const_step_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget) == hipSuccess,
                                    "march_rays_train_arena: cannot raise the dynamic LDS limit");
                        attr_set = true;
                    }
                }
            }
            march_const_step_kernel<<<dim3(ceil_div(N, kCsRays)), dim3(kCsRays * 64u), lds, as_stream(stream)>>>(
                rays_o, rays_d, grid, lds ? occ_index : nullptr, cap, bound, contract != 0, max_steps, N, C, H, nears, fars,
                noises, chain_cap, rays, t_scratch, counter);
        } else {
            march_classify_kernel<<<dim3(ceil_div(chain_cap, 256u), min(N, 65535u)), dim3(256), 0, as_stream(stream)>>>(
                rays_o, rays_d, grid, bound, contract != 0, dt_gamma, max_steps, N, C, H, chain_cap, chain, chain_len,
                chain_code);
            march_walk_kernel<<<dim3(ceil_div(N, 4u)), dim3(256), 0, as_stream(stream)>>>(chain, chain_len, chain_code, N,
                                                                                         chain_cap, max_steps, rays, t_scratch);
        }
    } else if (occ_index) {
        NGP_REQUIRE(n_bits % 2048u == 0 && ((uintptr_t)grid & 7u) == 0 && ((uintptr_t)occ_index & 7u) == 0,
                    "march_rays_train_arena: occupancy index needs C*H^3 to be a multiple of 2048 and 8-byte aligned buffers");
        const uint32_t n_groups = (uint32_t)(n_bits / 2048u);
        // 96 KiB, not all 160: the march runs beside the main stream's kernels and must leave room for their
        // workgroups on its CUs (a persistent-grid kernel that cannot place a workgroup there runs a second round)
        constexpr uint32_t kLdsBudget = 96u * 1024u;
        NGP_REQUIRE((size_t)n_groups * 8 + 8 * 1024 <= kLdsBudget, "march_rays_train_arena: too many cascades for the LDS index");
        const uint32_t cap = (kLdsBudget - n_groups * 8u) / 8u;
        static bool attr_set = false;
        if (!attr_set) {
            NGP_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void *>(march_count_indexed_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget) == hipSuccess,
                        "march_rays_train_arena: cannot raise the dynamic LDS limit");
            attr_set = true;
        }
        march_count_indexed_kernel<<<dim3(ceil_div(N, kFastBlock)), dim3(kFastBlock), kLdsBudget, as_stream(stream)>>>(
            rays_o, rays_d, grid, occ_index, cap, bound, contract != 0, dt_gamma, max_steps, N, C, H, nears, fars, noises,
            rays, t_scratch);
    } else {
        march_count_kernel<1><<<NGP_1D(N, kRayBlock)>>>(rays_o, rays_d, grid, bound, contract != 0, dt_gamma, max_steps, N,
                                                       C, H, nears, fars, noises, rays, t_scratch);
    }
    march_scan_kernel<<<dim3(1), dim3(1024), 0, as_stream(stream)>>>(rays, N, counter, M_cap, true);
    march_expand_kernel<<<dim3(ceil_div(N, 4u)), dim3(256), 0, as_stream(stream)>>>(
        rays_o, rays_d, rays_ldir, bound, contract != 0, dt_gamma, max_steps, N, C, H, rays, t_scratch, xyzs, dirs, ts,
        rays_ldir ? ldirs : nullptr, ray_idx);
    NGP_CHECK_LAUNCH("march_rays_train_arena");
    return NGP_OK;
}

extern "C" int ngp_x_march_rays_train_arena(const float *rays_o, const float *rays_d, const float *rays_ldir,
                                            const uint8_t *grid, float bound, int contract, float dt_gamma,
                                            uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                            const float *nears, const float *fars, const float *noises,
                                            float *t_scratch, uint32_t M_cap, float *xyzs, float *dirs, float *ts,
                                            float *ldirs, int32_t *rays, int32_t *counter, int32_t *ray_idx,
                                            const uint32_t *occ_index, float *chain, uint16_t *chain_code,
                                            int32_t *chain_len, uint32_t chain_cap, ngp_stream_t stream)
{
    return ngp_x_march_rays_train_arena_stage(rays_o, rays_d, rays_ldir, grid, bound, contract, dt_gamma, max_steps, N, C, H,
                                              nears, fars, noises, t_scratch, M_cap, xyzs, dirs, ts, ldirs, rays, counter,
                                              ray_idx, occ_index, chain, chain_code, chain_len, chain_cap, 0, stream);
}

extern "C" size_t ngp_x_occupancy_index_bytes(uint32_t C, uint32_t H)
{
    const uint64_t n_words64 = (uint64_t)C * H * H * H / 64u;
    return (size_t)(kOccHeader * 4 + (n_words64 / 32u) * 8 + n_words64 * 8);
}

extern "C" int ngp_x_build_occupancy_index(const uint8_t *grid, uint32_t C, uint32_t H, uint32_t *index,
                                           ngp_stream_t stream)
{
    NGP_REQUIRE(grid && index, "build_occupancy_index: null tensor");
    const uint64_t n_bits = (uint64_t)C * H * H * H;
    NGP_REQUIRE(n_bits > 0 && n_bits % 2048u == 0 && n_bits / 64u < (1ull << 31),
                "build_occupancy_index: C*H^3 must be a positive multiple of 2048");
    NGP_REQUIRE(((uintptr_t)grid & 7u) == 0 && ((uintptr_t)index & 7u) == 0, "build_occupancy_index: buffers must be 8-byte aligned");
    occupancy_index_kernel<<<dim3(1), dim3(1024), 0, as_stream(stream)>>>(grid, (uint32_t)(n_bits / 64u), index);
    NGP_CHECK_LAUNCH("build_occupancy_index");
    return NGP_OK;
}

extern "C" int ngp_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *ts,
                                                const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                                float *weights, float *weights_sum, float *depth, float *image,
                                                ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays && weights_sum && depth && image, "composite_rays_train_forward: null tensor");
    NGP_REQUIRE(M == 0 || (sigmas && rgbs && ts && weights), "composite_rays_train_forward: null sample tensor");
    composite_train_forward_kernel<<<NGP_1D(N, kRayBlock)>>>(sigmas, rgbs, ts, rays, M, N, T_thresh, weights, weights_sum,
                                                            depth, image);
    NGP_CHECK_LAUNCH("composite_rays_train_forward");
    return NGP_OK;
}

extern "C" int ngp_composite_rays_train_backward(const float *grad_weights, const float *grad_weights_sum,
                                                 const float *grad_depth, const float *grad_image,
                                                 const float *sigmas, const float *rgbs, const float *ts,
                                                 const int32_t *rays, const float *weights_sum, const float *depth,
                                                 const float *image, uint32_t M, uint32_t N, float T_thresh,
                                                 float *grad_sigmas, float *grad_rgbs, ngp_stream_t stream)
{
    if (N == 0 || M == 0) return NGP_OK;
    NGP_REQUIRE(grad_weights_sum && grad_depth && grad_image && rays && weights_sum && depth && image,
                "composite_rays_train_backward: null tensor");
    NGP_REQUIRE(grad_weights && sigmas && rgbs && ts && grad_sigmas && grad_rgbs,
                "composite_rays_train_backward: null sample tensor");
    composite_train_backward_kernel<<<NGP_1D(N, kRayBlock)>>>(grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas,
                                                             rgbs, ts, rays, weights_sum, depth, image, M, N, T_thresh,
                                                             grad_sigmas, grad_rgbs);
    NGP_CHECK_LAUNCH("composite_rays_train_backward");
    return NGP_OK;
}

extern "C" int ngp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                              const float *rays_o, const float *rays_d, float bound, int contract, float dt_gamma,
                              uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid, const float *nears,
                              const float *fars, float *xyzs, float *dirs, float *ts, const float *noises,
                              ngp_stream_t stream)
{
    if (n_alive == 0 || n_step == 0) return NGP_OK;
    (void)nears;
    NGP_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && ts && noises,
                "march_rays: null tensor");
    march_rays_kernel<<<NGP_1D(n_alive, kRayBlock)>>>(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound,
                                                     contract != 0, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, ts,
                                                     noises);
    NGP_CHECK_LAUNCH("march_rays");
    return NGP_OK;
}

extern "C" int ngp_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                                  float *rays_t, const float *sigmas, const float *rgbs, const float *ts,
                                  float *weights_sum, float *depth, float *image, ngp_stream_t stream)
{
    if (n_alive == 0) return NGP_OK;
    NGP_REQUIRE(rays_alive && rays_t && sigmas && rgbs && ts && weights_sum && depth && image, "composite_rays: null tensor");
    composite_rays_kernel<<<NGP_1D(n_alive, kRayBlock)>>>(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, ts,
                                                         weights_sum, depth, image);
    NGP_CHECK_LAUNCH("composite_rays");
    return NGP_OK;
}

extern "C" int ngp_x_march_rays_train_backward(const float *grad_xyzs, const float *grad_dirs, const float *ts,
                                               const int32_t *rays, uint32_t N, uint32_t M, float *grad_rays_o,
                                               float *grad_rays_d, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays && grad_rays_o && grad_rays_d, "march_rays_train_backward: null tensor");
    NGP_REQUIRE(M == 0 || (grad_xyzs && ts), "march_rays_train_backward: null sample tensor");
    march_train_backward_kernel<<<dim3(ceil_div(N, 4u)), dim3(256), 0, as_stream(stream)>>>(grad_xyzs, grad_dirs, ts, rays, N,
                                                                                          M, grad_rays_o, grad_rays_d);
    NGP_CHECK_LAUNCH("march_rays_train_backward");
    return NGP_OK;
}
