// Kernels of the fused training step (raw_ngp_amd/nerf/engine.py): the same arithmetic as the reference's
// per-op path, arranged so that one optimiser step needs ~20 launches, no host synchronisation and no
// intermediate layout copies.  Every kernel reads the number of live samples from device memory (the arena
// march's counter), so launch geometry is fixed by the arena capacity and the step can be graph-captured.
//
//   grid_forward_slab     gridencoder.cu:82-249 on world-space points: folds GridEncoder.forward's
//                         (x + bound) / (2 bound) map (grid.py:161) into the load and writes the [L, stride, 2]
//                         slab the fused MLP consumes
//   composite_* (wave)    raymarching.cu:519-597 / :623-712 with one WAVE per ray: transmittance by a
//                         wave-wide exclusive prefix product, running sums by wave scans (the reference walks
//                         each ray serially from one lane); the loss variant also folds in the harness'
//                         MSE against gt*alpha + bg*(1-alpha) (train_utils.py:503-541) and bg mixing
//                         (renderer.py:672), producing d sigma / d rgb directly
//   adam                  torch.optim.Adam (main.py:245: eps 1e-15, no weight decay) in one pass
//   near_far_v2           the torch slab test run_cuda really uses (renderer.py:139-158): /(d + 1e-15), miss -> 1e9
#include "binned_common.hpp"
#include "rng_common.hpp"

namespace ngp {

// ------------------------------------------------------------------ hash-grid forward into the slab
// COUNT: also size the bins of the binned table backward (grid_backward_binned.hip: what bin_count_kernel does) --
// the rows of all 8 corners are in registers here and the kernel waits on its gathers anyway, so the LDS histogram
// rides along for free and the backward needs no counting pass of its own.
// JAC: also write d out / d x01 (gridencoder.cu:205-247) as a level-major slab dydx[level][stride][3][2] -- what the ray
// gradients of pose refinement contract with d enc (pose_kernels.hip: ray_gradients)
template <bool COUNT, bool JAC = false, bool PAIR = true>
__global__ __launch_bounds__(kBlock) void grid_forward_slab_kernel(
    const float *__restrict__ xyzs, float bound, const float *__restrict__ table, const int32_t *__restrict__ offsets,
    float *__restrict__ out, float *__restrict__ inputs01, const int32_t *__restrict__ B_dev, uint32_t B_cap,
    uint32_t stride, uint32_t nchunks, LevelRes lv, uint32_t gridtype, bool align_corners, uint32_t interp, WsLayout w,
    float *__restrict__ dydx = nullptr, uint32_t snake_levels = 0, uint32_t placed_levels = 0, LevelPlacement place = {},
    uint32_t level_lo = 0, uint32_t persistent = 0)
{
    // level_lo: the launch covers levels level_lo .. level_lo + (levels of the grid) - 1 (ngp_x_grid_encode_forward_slab_levels)
    extern __shared__ uint32_t hist[];
    // one work item = 256 points of one level (the body keeps the indentation it had as the kernel's own)
    auto do_tile = [&](uint32_t level, uint32_t tile_) {
    const uint32_t b0 = tile_ * kBlock, b = b0 + threadIdx.x;
    const uint32_t B = B_dev ? min((uint32_t)max(B_dev[0], 0), B_cap) : B_cap;
    if (b0 >= B) return;   // whole workgroup
    const Geom<3> g = make_geom<3>(offsets, level, lv.res[level], gridtype);
    uint32_t first = 0, nbins = 0;
    if (COUNT) {
        first = w.chunk_base[level];
        nbins = w.chunk_base[level + 1] - first;
        for (uint32_t i = threadIdx.x; i < nbins; i += kBlock) hist[i] = 0;
        __syncthreads();
    }
    const float *__restrict__ tab = table + (size_t)(uint32_t)offsets[level] * 2;
    const bool have = b < B;
    float x[3] = {0.f, 0.f, 0.f};
    if (have) {
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) x[d] = (xyzs[(size_t)b * 3 + d] + bound) / (2.0f * bound);
        if (level == 0 && inputs01) {
#pragma unroll
            for (uint32_t d = 0; d < 3; d++) inputs01[(size_t)b * 3 + d] = x[d];
        }
    }
    float2 *dst = reinterpret_cast<float2 *>(out) + (size_t)level * stride + b;
    Cell<3> cl = {};
    const bool live = have && locate<3>(x, g.res, align_corners, interp, cl);
    // The gathers are bound by the address pipeline (one lane per clock), not by bytes: fetch the two x-neighbours of a
    // corner pair with ONE 16-byte load whenever their rows are adjacent -- always on dense levels (stride 1 along x),
    // and on hashed levels when the cell's x is even (prime_x = 1, so the two hashes differ in bit 0 only).
    // (PAIR = false, plain 8-byte gathers: 48 us against 46 in the step -- although in tools/ubench/gather_lines.hip, where
    // nothing but the loads is left, the unpaired stream is the faster one: 36 us against 46.)
    uint32_t row_id[8];
    float ax = 0.f, ay = 0.f;
    if (live) {
        Row<2> rows[8];
        float wts[8];
        const AxisTerms<3> terms = axis_terms<3>(g, cl);
#pragma unroll
        for (uint32_t yz = 0; yz < 4; yz++) {
            const uint32_t ra = row_from_terms<3>(g, terms, yz * 2u), rb = row_from_terms<3>(g, terms, yz * 2u + 1u);
            // weights in the reference's multiplication order: x factor first
            float wa = 1.0f - cl.f[0], wb = cl.f[0];
#pragma unroll
            for (uint32_t d = 1; d < 3; d++) {
                const float f = (yz & (1u << (d - 1))) ? cl.f[d] : 1.0f - cl.f[d];
                wa *= f;
                wb *= f;
            }
            const uint32_t ca = yz * 2u, cb = yz * 2u + 1u;
            wts[ca] = wa;
            wts[cb] = wb;
            row_id[ca] = ra;
            row_id[cb] = rb;
            if (PAIR && (rb == ra + 1u || ra == rb + 1u)) {
                const uint32_t lo = min(ra, rb);
                const float4 v = *reinterpret_cast<const float4 *>(tab + (size_t)lo * 2);
                const bool a_first = ra < rb;
                rows[ca].v[0] = a_first ? v.x : v.z;
                rows[ca].v[1] = a_first ? v.y : v.w;
                rows[cb].v[0] = a_first ? v.z : v.x;
                rows[cb].v[1] = a_first ? v.w : v.y;
            } else {
                rows[ca].load(tab + (size_t)ra * 2);
                rows[cb].load(tab + (size_t)rb * 2);
            }
        }
#pragma unroll
        for (uint32_t corner = 0; corner < 8; corner++) {
            ax = fmaf(wts[corner], rows[corner].v[0], ax);
            ay = fmaf(wts[corner], rows[corner].v[1], ay);
        }
        if constexpr (JAC) {   // corner bit d set = upper neighbour along d; same expression order as gridencoder.hip
            const float scale = (float)(align_corners ? g.res - 1u : g.res);
            float2 *jd = reinterpret_cast<float2 *>(dydx) + ((size_t)level * stride + b) * 3;
#pragma unroll
            for (uint32_t gd = 0; gd < 3; gd++) {
                float g0 = 0.f, g1 = 0.f;
#pragma unroll
                for (uint32_t combo = 0; combo < 4; combo++) {
                    float wj = scale;
                    uint32_t lo = 0;
#pragma unroll
                    for (uint32_t nd = 0; nd < 2; nd++) {
                        const uint32_t d = nd >= gd ? nd + 1 : nd;
                        if (combo & (1u << nd)) {
                            wj *= cl.f[d];
                            lo |= 1u << d;
                        } else {
                            wj *= 1.0f - cl.f[d];
                        }
                    }
                    const uint32_t hi = lo | (1u << gd);
                    g0 = fmaf(wj * (rows[hi].v[0] - rows[lo].v[0]), cl.df[gd], g0);
                    g1 = fmaf(wj * (rows[hi].v[1] - rows[lo].v[1]), cl.df[gd], g1);
                }
                jd[gd] = make_float2(g0, g1);
            }
        }
    }
    if (have) *dst = make_float2(ax, ay);   // zeros outside [0,1]^3, like the reference
    if constexpr (JAC) {
        if (have && !live) {
            float2 *jd = reinterpret_cast<float2 *>(dydx) + ((size_t)level * stride + b) * 3;
            jd[0] = jd[1] = jd[2] = make_float2(0.f, 0.f);
        }
    }
    if (COUNT) {
        bool emit = live;
        if (mergeable(g, w)) {   // level-uniform; every lane of the wave gets here
            uint32_t dist;
            emit = run_shape(live ? cell_key(cl) : kDeadKey, live, dist);
        }
        if (emit) {
#pragma unroll
            for (uint32_t corner = 0; corner < 8; corner++) atomicAdd(&hist[row_id[corner] >> kChunkShift], 1u);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nbins; i += kBlock)
            if (hist[i]) atomicAdd(&w.count[first + i], hist[i]);
    }
    };
    if (!COUNT && persistent) {
        // Persistent form (the number of points is only known on the device: a grid sized for the arena's capacity is four
        // fifths workgroups that find nothing to do, dealt to the XCDs in between the ones that do).  gridDim.x = 8 x (workgroups
        // per XCD); the workgroups with the same blockIdx % 8 share an XCD and stride through that XCD's list (snake: level k,
        // then level 15 - k) -- every XCD at its own pace, and with fewer workgroups per CU than fit, so that the march on the
        // side stream keeps its wave slots: forward 43.8 -> 41.3 us in the step, 0.2855 -> 0.2773 ms/step (kSlabWgsPerCu).
        const uint32_t k = blockIdx.x & 7u, G8 = gridDim.x >> 3;
        const uint32_t Bq = B_dev ? min((uint32_t)max(B_dev[0], 0), B_cap) : B_cap;
        const uint32_t tiles = (Bq + kBlock - 1u) / kBlock, len = snake_rounds(snake_levels) * tiles;
        // (static stride, no work counter: the items of a level cost the same; a counter per XCD was tried -- one atomic per item
        // costs more than the imbalance it removes)
        for (uint32_t it = blockIdx.x >> 3; it < len; it += G8) {
            const uint32_t round = it / tiles, t = it - round * tiles;
            const uint32_t lvl = (round >> 1) * 16u + ((round & 1u) ? 15u - k : k);
            if (lvl < snake_levels) do_tile(lvl + level_lo, t);
        }
        return;
    }
    uint32_t level, tile_;
    if (placed_levels) {
        placed_level_tile(place, blockIdx.x, nchunks, placed_levels, level, tile_);
        if (level == kNoLevel) return;
    } else if (snake_levels) {
        snake_level_tile(blockIdx.x, nchunks, snake_levels, level, tile_);
        if (level == kNoLevel) return;
    } else {
        const uint32_t item = xcd_remap(blockIdx.x, gridDim.x);
        level = item / nchunks;
        tile_ = item - level * nchunks;
    }
    level += level_lo;
    do_tile(level, tile_);
}

// ------------------------------------------------------------------ wave scans
__device__ __forceinline__ float wave_excl_prod(float v, uint32_t lane)
{   // exclusive prefix product over the 64 lanes
    float inc = v;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const float up = __shfl_up(inc, d, 64);
        if (lane >= d) inc *= up;
    }
    const float ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 1.0f : ex;
}
__device__ __forceinline__ float wave_incl_sum(float v, uint32_t lane)
{
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const float up = __shfl_up(v, d, 64);
        if (lane >= d) v += up;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ------------------------------------------------------------------ compositing, one wave per ray
// writes weights for every sample of the ray (0 after the early stop), so the caller need not pre-zero
__global__ __launch_bounds__(256) void composite_forward_wave_kernel(
    const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ ts,
    const int32_t *__restrict__ rays, uint32_t M, uint32_t N, float T_thresh, float *__restrict__ weights,
    float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image)
{
    const uint32_t n = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[(size_t)n * 2], cnt = (uint32_t)rays[(size_t)n * 2 + 1];
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
    bool stopped = false;
    if (cnt != 0 && off + cnt <= M) {
        for (uint32_t base = 0; base < cnt; base += 64u) {
            const uint32_t i = off + base + lane;
            const bool have = base + lane < cnt;
            if (stopped) {   // wave-uniform
                if (have) weights[i] = 0.0f;
                continue;
            }
            float alpha = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, t = 0.f;
            if (have) {
                const float2 tt = reinterpret_cast<const float2 *>(ts)[i];
                alpha = 1.0f - __expf(-sigmas[i] * tt.y);
                t = tt.x;
                c0 = rgbs[(size_t)i * 3];
                c1 = rgbs[(size_t)i * 3 + 1];
                c2 = rgbs[(size_t)i * 3 + 2];
            }
            const float Tb = T * wave_excl_prod(1.0f - alpha, lane);   // transmittance in front of the sample
            const float Ta = Tb * (1.0f - alpha);
            // the reference keeps the sample that drives T below the threshold and drops everything after it
            const unsigned long long hit = __ballot(have && Ta < T_thresh);
            const uint32_t last = hit ? (uint32_t)__ffsll((long long)hit) - 1u : 63u;
            const bool use = have && lane <= last;
            const float w = use ? alpha * Tb : 0.0f;
            if (have) weights[i] = w;
            r += wave_sum(w * c0);
            g += wave_sum(w * c1);
            b += wave_sum(w * c2);
            ws += wave_sum(w);
            d += wave_sum(w * t);
            T = __shfl(Ta, min(last, 63u), 64);
            stopped = hit != 0ull;
        }
    }
    if (lane == 0) {
        weights_sum[n] = ws;
        depth[n] = d;
        image[(size_t)n * 3] = r;
        image[(size_t)n * 3 + 1] = g;
        image[(size_t)n * 3 + 2] = b;
    }
}

// MODE 0: plain backward with caller-provided gradients (same contract as ngp_composite_rays_train_backward,
//         but every sample of the ray is written: zeros after the early stop)
// MODE 1: gradients of the MSE loss  mean_{n,c} ((image + (1 - ws) bg - gt)^2)  with gt = rgb*a + bg*(1-a);
//         also accumulates the loss value into loss_out[0]
// MODE 2: MODE 1 with the forward pass inside: each wave first composites its ray (the arithmetic of
//         composite_forward_wave_kernel, so the totals are the same bits), writes weights_sum / depth / image and goes
//         straight into the backward -- the training step needs no separate forward launch
constexpr uint32_t kCompBwdBlock = 1024;   // 16 rays per workgroup: one loss atomic per workgroup (same-address
                                           // global atomics serialise: 4096 of them cost ~60 us)
// The RawNeRF-style loss of the HDR mode (train_utils.py:512-536): the prediction is scaled by the ray's exposure and
// clipped at 1, the squared residual is weighted by the squared gradient of the log tone curve 1 / (1e-3 + sg(clip)),
//     loss = sum(resid^2 * scaling^2 * lossmult * loss_weight) / sum(lossmult)
// exposure == NULL selects the MSE.  weight: optional [N,3] = lossmult * loss_weight (NULL = 1), inv_norm = 1 / sum(lossmult)
struct HdrLoss {
    const float *exposure = nullptr;
    const float *weight = nullptr;
    float inv_norm = 0.0f;
    // adaptive ray batches (train_utils.py:563-564): only the first n_live[0] of the N ray slots carry rays; the loss is
    // the mean over those (NULL: all N)
    const int32_t *n_live = nullptr;
    // entropy of the accumulated opacity (train_utils.py:554-557): loss += lambda * mean_rays(H(clamp(ws, 1e-5, 1 - 1e-5))),
    // H(w) = -w log2 w - (1 - w) log2(1 - w); its gradient joins d loss / d weights_sum
    float lambda_entropy = 0.0f;
    // MODE 2: the number of samples in front of (and including) the compositor's early stop, per ray slot -- the samples
    // that can have a gradient at all (live_index_kernel turns the counts into a compact list for the backward kernels)
    int32_t *live_out = nullptr;
    // MODE 2: a term over the samples' compositing weights, loss += lambda_sample * sum_i w_i sample_term[i] (the
    // orientation term, renderer.py:571: `weights` reach the loss directly).  Its gradient is the reference's, i.e. what
    // kernel_composite_rays_train_backward does with grad_weights (raymarching.cu:694): added to grad_weights_sum at the
    // sample itself
    const float *sample_term = nullptr;
    float lambda_sample = 0.0f;
    float *term_weight_out = nullptr;   // optional [M]: lambda_sample * weights[i] (0 behind the early stop), for consumers of
                                        // the term's other inputs (ngp_x_ray_gradients_terms: the view direction)
};

template <int MODE>
__global__ __launch_bounds__(kCompBwdBlock) void composite_backward_wave_kernel(
    const float *__restrict__ grad_weights, const float *__restrict__ grad_weights_sum,
    const float *__restrict__ grad_depth, const float *__restrict__ grad_image, const float *__restrict__ gt_rgba,
    const float *__restrict__ bg_rgb, float bg_const, const float *__restrict__ sigmas, const float *__restrict__ rgbs,
    const float *__restrict__ ts, const int32_t *__restrict__ rays, const float *weights_sum, const float *depth,
    const float *image, uint32_t M, uint32_t N, float T_thresh, float *__restrict__ grad_sigmas,
    float *__restrict__ grad_rgbs, float *__restrict__ loss_out, float *ws_out, float *depth_out, float *image_out,
    HdrLoss hdr = HdrLoss{})
{
    __shared__ float ray_err[kCompBwdBlock / 64];
    const uint32_t n = (blockIdx.x * kCompBwdBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (MODE >= 1) {   // the whole workgroup passes the barrier before any wave leaves
        if (lane == 0) ray_err[threadIdx.x >> 6] = 0.0f;
    }
    const uint32_t n_rays = hdr.n_live ? min((uint32_t)max(hdr.n_live[0], 1), N) : N;
    const bool in_range = n < n_rays;
    const uint32_t nn = in_range ? n : 0u;
    const uint32_t off = (uint32_t)rays[(size_t)nn * 2], cnt = (uint32_t)rays[(size_t)nn * 2 + 1];
    const bool live = in_range && cnt != 0 && off + cnt <= M;
    float rF, gF, bF, wsF, dF, term_sum = 0.0f;
    if (MODE == 2) {
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
        uint32_t used = live ? cnt : 0u;   // samples up to and including the early stop
        if (live) {
            for (uint32_t base = 0; base < cnt; base += 64u) {
                const uint32_t i = off + base + lane;
                const bool have = base + lane < cnt;
                float alpha = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, t = 0.f, term = 0.f;
                if (have) {
                    const float2 tt = reinterpret_cast<const float2 *>(ts)[i];
                    alpha = 1.0f - __expf(-sigmas[i] * tt.y);
                    t = tt.x;
                    c0 = rgbs[(size_t)i * 3];
                    c1 = rgbs[(size_t)i * 3 + 1];
                    c2 = rgbs[(size_t)i * 3 + 2];
                    if (hdr.sample_term) term = hdr.sample_term[i];
                }
                const float Tb = T * wave_excl_prod(1.0f - alpha, lane);
                const float Ta = Tb * (1.0f - alpha);
                const unsigned long long hit = __ballot(have && Ta < T_thresh);
                const uint32_t last = hit ? (uint32_t)__ffsll((long long)hit) - 1u : 63u;
                const float w = (have && lane <= last) ? alpha * Tb : 0.0f;
                r += wave_sum(w * c0);
                g += wave_sum(w * c1);
                b += wave_sum(w * c2);
                ws += wave_sum(w);
                d += wave_sum(w * t);
                if (hdr.sample_term) term_sum += wave_sum(w * term);
                T = __shfl(Ta, min(last, 63u), 64);
                if (hit != 0ull) {   // wave-uniform
                    used = base + last + 1u;
                    break;
                }
            }
        }
        if (hdr.live_out && lane == 0 && n < N) hdr.live_out[n] = (int32_t)used;
        rF = r, gF = g, bF = b, wsF = ws, dF = d;
        if (lane == 0 && in_range) {
            ws_out[n] = ws;
            depth_out[n] = d;
            image_out[(size_t)n * 3] = r;
            image_out[(size_t)n * 3 + 1] = g;
            image_out[(size_t)n * 3 + 2] = b;
        }
    } else {
        rF = image[(size_t)nn * 3], gF = image[(size_t)nn * 3 + 1], bF = image[(size_t)nn * 3 + 2];
        wsF = weights_sum[nn], dF = depth[nn];
    }
    float gr, gg, gb, gws, gd;
    if (MODE >= 1) {
        const float4 px = reinterpret_cast<const float4 *>(gt_rgba)[nn];
        const float b0 = bg_rgb ? bg_rgb[(size_t)nn * 3] : bg_const, b1 = bg_rgb ? bg_rgb[(size_t)nn * 3 + 1] : bg_const,
                    b2 = bg_rgb ? bg_rgb[(size_t)nn * 3 + 2] : bg_const;
        float ray_loss;
        if (hdr.exposure) {
            const float ex = hdr.exposure[nn];
            const float pred[3] = {rF + (1.0f - wsF) * b0, gF + (1.0f - wsF) * b1, bF + (1.0f - wsF) * b2};
            const float gt[3] = {px.x * px.w + b0 * (1.0f - px.w), px.y * px.w + b1 * (1.0f - px.w),
                                 px.z * px.w + b2 * (1.0f - px.w)};
            float gch[3];
            ray_loss = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float scaled = pred[c] * ex, clip = fminf(1.0f, scaled);
                const float sg = 1.0f / (1e-3f + clip), resid = clip - gt[c];
                const float wgt = (hdr.weight ? hdr.weight[(size_t)nn * 3 + c] : 1.0f) *
                                  (hdr.n_live ? 1.0f / (3.0f * (float)n_rays) : hdr.inv_norm);
                ray_loss += resid * resid * (sg * sg) * wgt;
                gch[c] = scaled < 1.0f ? 2.0f * resid * (sg * sg) * wgt * ex : 0.0f;   // no gradient through the clip
            }
            gr = gch[0];
            gg = gch[1];
            gb = gch[2];
        } else {
            const float e0 = (rF + (1.0f - wsF) * b0) - (px.x * px.w + b0 * (1.0f - px.w));
            const float e1 = (gF + (1.0f - wsF) * b1) - (px.y * px.w + b1 * (1.0f - px.w));
            const float e2 = (bF + (1.0f - wsF) * b2) - (px.z * px.w + b2 * (1.0f - px.w));
            const float k = 2.0f / (3.0f * (float)n_rays);
            gr = k * e0;
            gg = k * e1;
            gb = k * e2;
            ray_loss = (e0 * e0 + e1 * e1 + e2 * e2) / (3.0f * (float)n_rays);
        }
        gws = -(gr * b0 + gg * b1 + gb * b2);
        gd = 0.0f;
        if (hdr.lambda_entropy > 0.0f) {
            const float wcl = fminf(fmaxf(wsF, 1e-5f), 1.0f - 1e-5f), kk = hdr.lambda_entropy / (float)n_rays;
            ray_loss += kk * (-wcl * log2f(wcl) - (1.0f - wcl) * log2f(1.0f - wcl));
            // (torch.clamp passes the gradient on [min, max], ends included)
            if (wsF >= 1e-5f && wsF <= 1.0f - 1e-5f) gws += kk * (log2f(1.0f - wcl) - log2f(wcl));
        }
        if (MODE == 2) ray_loss += hdr.lambda_sample * term_sum;   // (a SUM over the samples: torch.mean of a scalar)
        if (lane == 0 && in_range) ray_err[threadIdx.x >> 6] = ray_loss;
        __syncthreads();
        if (threadIdx.x == 0) {
            float sum = 0.0f;
#pragma unroll
            for (uint32_t k = 0; k < kCompBwdBlock / 64; k++) sum += ray_err[k];
            atomicAdd(loss_out, sum);
        }
    } else {
        gr = grad_image[(size_t)nn * 3];
        gg = grad_image[(size_t)nn * 3 + 1];
        gb = grad_image[(size_t)nn * 3 + 2];
        gws = grad_weights_sum[nn];
        gd = grad_depth[nn];
    }
    if (!live) return;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
    bool stopped = false;
    for (uint32_t base = 0; base < cnt; base += 64u) {
        const uint32_t i = off + base + lane;
        const bool have = base + lane < cnt;
        if (stopped) {
            if (have) {
                grad_sigmas[i] = 0.0f;
                grad_rgbs[(size_t)i * 3] = grad_rgbs[(size_t)i * 3 + 1] = grad_rgbs[(size_t)i * 3 + 2] = 0.0f;
                if (MODE == 2 && hdr.term_weight_out) hdr.term_weight_out[i] = 0.0f;
            }
            continue;
        }
        float alpha = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, t = 0.f, dt = 0.f, gw = 0.f;
        if (have) {
            const float2 tt = reinterpret_cast<const float2 *>(ts)[i];
            t = tt.x;
            dt = tt.y;
            alpha = 1.0f - __expf(-sigmas[i] * dt);
            c0 = rgbs[(size_t)i * 3];
            c1 = rgbs[(size_t)i * 3 + 1];
            c2 = rgbs[(size_t)i * 3 + 2];
            if (MODE == 0) gw = grad_weights[i];
            if (MODE == 2 && hdr.sample_term) gw = hdr.lambda_sample * hdr.sample_term[i];
        }
        const float Tb = T * wave_excl_prod(1.0f - alpha, lane);
        const float Ta = Tb * (1.0f - alpha);
        const unsigned long long hit = __ballot(have && Ta < T_thresh);
        const uint32_t last = hit ? (uint32_t)__ffsll((long long)hit) - 1u : 63u;
        const bool use = have && lane <= last;
        const float w = use ? alpha * Tb : 0.0f;
        // running sums up to and including this sample
        const float ri = r + wave_incl_sum(w * c0, lane), gi = g + wave_incl_sum(w * c1, lane),
                    bi = b + wave_incl_sum(w * c2, lane), wi = ws + wave_incl_sum(w, lane),
                    di = d + wave_incl_sum(w * t, lane);
        if (have) {
            float s = 0.0f, q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
            if (use) {
                q0 = gr * w;
                q1 = gg * w;
                q2 = gb * w;
                s = gr * fmaf(Ta, c0, -(rF - ri));
                s = fmaf(gg, fmaf(Ta, c1, -(gF - gi)), s);
                s = fmaf(gb, fmaf(Ta, c2, -(bF - bi)), s);
                s = fmaf(gws + gw, Ta - (wsF - wi), s);
                s = fmaf(gd, fmaf(Ta, t, -(dF - di)), s);
                s *= dt;
            }
            grad_sigmas[i] = s;
            if (MODE == 2 && hdr.term_weight_out) hdr.term_weight_out[i] = hdr.lambda_sample * w;
            grad_rgbs[(size_t)i * 3] = q0;
            grad_rgbs[(size_t)i * 3 + 1] = q1;
            grad_rgbs[(size_t)i * 3 + 2] = q2;
        }
        r = __shfl(ri, 63, 64);
        g = __shfl(gi, 63, 64);
        b = __shfl(bi, 63, 64);
        ws = __shfl(wi, 63, 64);
        d = __shfl(di, 63, 64);
        T = __shfl(Ta, min(last, 63u), 64);
        stopped = hit != 0ull;
    }
}

// ------------------------------------------------------------------ the samples that can have a gradient
// Behind the compositor's early stop (T < T_thresh) every sample of a ray has exactly zero output gradients -- a third of
// the batch late in training -- and ray tails do not line up with the 32-sample tiles of the MLP backward or the 512-sample
// tiles of the table backward's fill, which therefore carry them along.  This kernel turns the per-ray counts of samples in
// front of the stop (composite_backward_wave_kernel<2>: hdr.live_out) into a compact, ray-ordered list of sample indices;
// the backward kernels then run over the list (ngp_x_mlp_backward_dirs, ngp_x_grid_backward_binned_apply: sample_index).
// Same sums, fewer tiles.  One workgroup = 16 rays (a wave each); a workgroup adds up the counts of all rays in front of
// its own -- N integers out of L2 -- instead of waiting for its predecessors: no scan kernel, no look-back, any order.
__global__ __launch_bounds__(1024) void live_index_kernel(const int32_t *__restrict__ rays,
                                                         const int32_t *__restrict__ live_n, uint32_t N, uint32_t M_cap,
                                                         int32_t *__restrict__ live_idx, int32_t *__restrict__ live_count,
                                                         int32_t *__restrict__ live_off)
{
    __shared__ uint32_t s_part[16], s_mine[16];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u, r0 = blockIdx.x * 16u, n = r0 + w;
    uint32_t part = 0;
    for (uint32_t i = threadIdx.x; i < r0; i += 1024u) part += (uint32_t)live_n[i];
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    const uint32_t mine = n < N ? (uint32_t)live_n[n] : 0u;
    if (lane == 0) {
        s_part[w] = part;
        s_mine[w] = mine;
    }
    __syncthreads();
    uint32_t at = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        at += s_part[k] + (k < w ? s_mine[k] : 0u);
        total += s_part[k] + s_mine[k];
    }
    if (n < N) {
        const uint32_t off = (uint32_t)rays[(size_t)n * 2];
        for (uint32_t k = lane; k < mine && at + k < M_cap; k += 64u) live_idx[at + k] = (int32_t)(off + k);
        if (live_off && lane == 0) live_off[n] = (int32_t)at;   // where ray n's entries start (per-ray consumers)
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) live_count[0] = (int32_t)min(total, M_cap);
}

// ------------------------------------------------------------------ Adam
// torch.optim.Adam(step): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//                         p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),  bc_i = 1 - b_i^step
// G16: the gradient is stored as bfloat16 (the data-parallel wire format; never zeroed here)
template <bool G16 = false>
__device__ __forceinline__ void adam_span(float *__restrict__ p, const float *g, float *__restrict__ m,
                                          float *__restrict__ v, size_t n4, size_t n, float lr, float b1, float b2,
                                          float eps, float bc1, float rsqrt_bc2, bool zero_grad, float *g_mut, size_t tid,
                                          size_t nthreads)
{
    const float step_size = lr / bc1;
    const uint16_t *g16 = reinterpret_cast<const uint16_t *>(g);
    for (size_t i = tid; i < n4; i += nthreads) {
        float4 pp = reinterpret_cast<float4 *>(p)[i];
        float4 gg;
        if (G16) {
            const uint2 u = reinterpret_cast<const uint2 *>(g)[i];
            gg = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                             __uint_as_float(u.y & 0xffff0000u));
        } else {
            gg = reinterpret_cast<const float4 *>(g)[i];
        }
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
#define NGP_ADAM1(c)                                                       \
    mm.c = b1 * mm.c + (1.0f - b1) * gg.c;                                 \
    vv.c = b2 * vv.c + (1.0f - b2) * gg.c * gg.c;                          \
    pp.c -= step_size * (mm.c / (sqrtf(vv.c) * rsqrt_bc2 + eps));
        NGP_ADAM1(x) NGP_ADAM1(y) NGP_ADAM1(z) NGP_ADAM1(w)
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
        if (!G16 && zero_grad) reinterpret_cast<float4 *>(g_mut)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (size_t i = n4 * 4 + tid; i < n; i += nthreads) {   // tail
        const float gi = G16 ? __uint_as_float((uint32_t)g16[i] << 16) : g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi, vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) * rsqrt_bc2 + eps));
        if (!G16 && zero_grad) g_mut[i] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *g, float *__restrict__ m,
                                                   float *__restrict__ v, size_t n4, size_t n, float lr, float b1,
                                                   float b2, float eps, float bc1, float rsqrt_bc2, bool zero_grad,
                                                   float *g_mut, const float *__restrict__ hyper,
                                                   const uint32_t *__restrict__ skip)
{
    // dynamic loss scale: a step that saw a non-finite gradient takes no optimiser step (GradScaler.step, train_utils.py:897)
    if (skip && skip[0] != 0u) return;
    if (hyper) {   // {lr, 1 - beta1^t, 1/sqrt(1 - beta2^t)} written by schedule_kernel earlier on this stream
        lr = hyper[0];
        bc1 = hyper[1];
        rsqrt_bc2 = hyper[2];
    }
    adam_span(p, g, m, v, n4, n, lr, b1, b2, eps, bc1, rsqrt_bc2, zero_grad, g_mut, (size_t)blockIdx.x * 256 + threadIdx.x,
              (size_t)gridDim.x * 256);
}

struct AdamTensor {
    float *p, *g, *m, *v;
    size_t n;
    bool zero_grad;
};

// blocks [0, blocks_a) update tensor a, the rest tensor b
template <bool A16>
__global__ __launch_bounds__(256) void adam2_kernel(AdamTensor a, AdamTensor b, uint32_t blocks_a, float b1, float b2,
                                                    float eps, const float *__restrict__ hyper,
                                                    const uint32_t *__restrict__ skip)
{
    if (skip && skip[0] != 0u) return;   // (as adam_kernel)
    const bool first = blockIdx.x < blocks_a;
    const AdamTensor t = first ? a : b;
    const uint32_t blk = first ? blockIdx.x : blockIdx.x - blocks_a, nblk = first ? blocks_a : gridDim.x - blocks_a;
    if (A16 && first)
        adam_span<true>(t.p, t.g, t.m, t.v, t.n / 4, t.n, hyper[0], b1, b2, eps, hyper[1], hyper[2], false, t.g,
                        (size_t)blk * 256 + threadIdx.x, (size_t)nblk * 256);
    else
        adam_span(t.p, t.g, t.m, t.v, t.n / 4, t.n, hyper[0], b1, b2, eps, hyper[1], hyper[2], t.zero_grad, t.g,
                  (size_t)blk * 256 + threadIdx.x, (size_t)nblk * 256);
}

// ------------------------------------------------------------------ step state on the device
// The harness' scheduler (main.py:261: lr = lr0 * 0.1^min(step/iters, 1)) and Adam's bias corrections, kept on the
// device so that a captured step never needs a host-supplied scalar.
__global__ void schedule_kernel(uint32_t *step_counter, float *hyper, double lr0, double decay_steps, double b1, double b2)
{
    const uint32_t done = step_counter[0];
    const double t = (double)done + 1.0;
    const double frac = fmin((double)done / decay_steps, 1.0);
    hyper[0] = (float)(lr0 * pow(0.1, frac));
    hyper[1] = (float)(1.0 - pow(b1, t));
    hyper[2] = (float)(1.0 / sqrt(1.0 - pow(b2, t)));
    step_counter[0] = done + 1u;
}

__global__ void counter_add_kernel(uint32_t *counter, uint32_t delta) { counter[0] += delta; }

__global__ __launch_bounds__(1024) void step_begin_kernel(StepBegin a) { step_begin_block(a); }

// ------------------------------------------------------------------ ray batch sampling
// The random_image_batch collate + get_rays + target gather of the harness (nerf/provider.py, nerf/train_utils.py:96-172)
// as one kernel: every ray draws its own (view, pixel), builds its origin / direction from that view's pose and
// reads its target colour.  Pixel centre +0.5, camera looks down -z, y flipped, directions not normalised.
// Adaptive ray batches (`--adaptive_num_rays`, train_utils.py:563-564: num_rays <- round(num_points / samples * num_rays)
// after every step) without a host read: the batch drawn for step i + 1 looks at the sample count step i's march left on
// the device and at step i's ray count.  Only the first live[0] of the N ray slots get rays; the others are parked outside
// the volume (no samples, excluded from the loss).  prev_samples == NULL: no adaptation (all N rays live).
struct AdaptiveRays {
    const int32_t *prev_samples = nullptr;   // samples of the previous batch (arena counter)
    const int32_t *prev_live = nullptr;      // rays of the previous batch
    int32_t *live = nullptr;                 // out: rays of this batch
    uint32_t num_points = 0;                 // target samples per batch
    bool resolved = false;                   // live[0] was computed by adaptive_live_kernel (prev_live aliases live)
};

__device__ __forceinline__ uint32_t adaptive_live_count(const AdaptiveRays &ad, uint32_t N)
{   // python: int(round(num_points / num_points_seen * num_rays)), kept inside [1, N]
    const double ratio = (double)ad.num_points / (double)max(ad.prev_samples[0], 1);
    return (uint32_t)fmin(fmax(rint(ratio * (double)max(ad.prev_live[0], 1)), 1.0), (double)N);
}

// One ray slot (no prefetch, pose refinement): the previous batch's ray count and this batch's are the SAME device word.
// Every workgroup of the sampler reads it while thread 0 would overwrite it, so the new count is formed here, by one
// thread, in a launch of its own in front of the sampler, which then only reads it.
__global__ void adaptive_live_kernel(AdaptiveRays ad, uint32_t N) { ad.live[0] = (int32_t)adaptive_live_count(ad, N); }

__global__ __launch_bounds__(256) void sample_rays_kernel(
    const uint8_t *__restrict__ images, uint32_t V, uint32_t H, uint32_t W, uint32_t C, const float *__restrict__ poses,
    float fx, float fy, float cx, float cy, uint32_t N, uint32_t seed_lo, uint32_t seed_hi,
    const uint32_t *__restrict__ draw_dev, uint32_t draw, float *__restrict__ rays_o, float *__restrict__ rays_d,
    float *__restrict__ gt, float *__restrict__ noises, float *__restrict__ bg, int32_t *__restrict__ index,
    const float *__restrict__ view_ldirs = nullptr, float *__restrict__ rays_ldir = nullptr,
    AdaptiveRays ad = AdaptiveRays{}, const float *__restrict__ view_exposure = nullptr,
    float *__restrict__ exposure = nullptr)
{
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    if (ad.live) {
        uint32_t n_live = N;
        if (ad.resolved)
            n_live = (uint32_t)ad.live[0];
        else if (ad.prev_samples)
            n_live = adaptive_live_count(ad, N);
        if (n == 0 && !ad.resolved) ad.live[0] = (int32_t)n_live;
        if (n >= n_live) {       // a parked slot: the ray starts far outside the volume and points away from it
            rays_o[(size_t)n * 3] = rays_o[(size_t)n * 3 + 1] = 0.0f;
            rays_o[(size_t)n * 3 + 2] = 1e6f;
            rays_d[(size_t)n * 3] = rays_d[(size_t)n * 3 + 1] = 0.0f;
            rays_d[(size_t)n * 3 + 2] = 1.0f;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) gt[(size_t)n * 4 + k] = 0.0f;
            if (noises) noises[n] = 0.0f;
            if (bg) bg[(size_t)n * 3] = bg[(size_t)n * 3 + 1] = bg[(size_t)n * 3 + 2] = 0.0f;
            if (index) index[2 * n] = index[2 * n + 1] = -1;
            if (rays_ldir) {
                rays_ldir[(size_t)n * 3] = rays_ldir[(size_t)n * 3 + 1] = 0.0f;
                rays_ldir[(size_t)n * 3 + 2] = 1.0f;
            }
            if (exposure) exposure[n] = 1.0f;
            return;
        }
    }
    if (draw_dev) draw = draw_dev[0];
    uint32_t r[4] = {n, draw, 0u, 0u};
    philox4x32_10(r, seed_lo, seed_hi);
    const uint32_t view = __umulhi(r[0], V), pix = __umulhi(r[1], H * W);
    const uint32_t j = pix / W, i = pix - j * W;
    const float dx = ((float)i + 0.5f - cx) / fx, dy = -(((float)j + 0.5f - cy) / fy), dz = -1.0f;
    const float *P = poses + (size_t)view * 16;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        rays_d[(size_t)n * 3 + k] = dx * P[4 * k] + dy * P[4 * k + 1] + dz * P[4 * k + 2];
        rays_o[(size_t)n * 3 + k] = P[4 * k + 3];
    }
    const uint8_t *px = images + ((size_t)view * H * W + pix) * C;
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) gt[(size_t)n * 4 + k] = (float)px[k] / 255.0f;
    gt[(size_t)n * 4 + 3] = C == 4 ? (float)px[3] / 255.0f : 1.0f;
    if (noises) noises[n] = u01(r[2]);
    if (bg) {
        uint32_t q[4] = {n, draw, 1u, 0u};
        philox4x32_10(q, seed_lo, seed_hi);
#pragma unroll
        for (int k = 0; k < 3; k++) bg[(size_t)n * 3 + k] = u01(q[k]);
    }
    if (index) {
        index[2 * n] = (int32_t)view;
        index[2 * n + 1] = (int32_t)pix;
    }
    if (rays_ldir) {   // one light direction per view (colmap_provider.py:619-620: metadict['ldirs'][index])
#pragma unroll
        for (int k = 0; k < 3; k++) rays_ldir[(size_t)n * 3 + k] = view_ldirs[(size_t)view * 3 + k];
    }
    if (exposure) exposure[n] = view_exposure[view];   // the exposure of the ray's image (colmap_provider.py:605-606)
}

// ------------------------------------------------------------------ near / far (torch semantics of run_cuda)
__global__ void near_far_v2_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                   const float *__restrict__ aabb, uint32_t N, float min_near, float *__restrict__ nears,
                                   float *__restrict__ fars)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float near = -3.402823466e+38f, far = 3.402823466e+38f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float o = rays_o[(size_t)n * 3 + a], dd = rays_d[(size_t)n * 3 + a] + 1e-15f;
        const float t0 = (aabb[a] - o) / dd, t1 = (aabb[3 + a] - o) / dd;
        near = fmaxf(near, t0 < t1 ? t0 : t1);
        far = fminf(far, t0 > t1 ? t0 : t1);
    }
    if (far < near) near = far = 1e9f;
    nears[n] = fmaxf(near, min_near);
    fars[n] = far;
}

}  // namespace ngp

using namespace ngp;

// workgroups of the persistent slab forward: kSlabWgsPerCu per CU (NGP_SLAB_WGS overrides; 0: the capacity-sized grid).  Six of
// the eight that fit: 5 / 6 / 7 / 8 per CU give 0.2787 / 0.2773 / 0.2822 / 0.2867 ms/step -- a full CU starves the side stream
constexpr uint32_t kSlabWgsPerCu = 6;
static uint32_t slab_persistent_blocks(bool eligible)
{
    static const uint32_t per_cu = getenv("NGP_SLAB_WGS") ? (uint32_t)atoi(getenv("NGP_SLAB_WGS")) : kSlabWgsPerCu;
    return eligible ? 256u * min(per_cu, 16u) : 0u;
}

extern "C" int ngp_x_grid_encode_forward_slab(const float *xyzs, float bound, const float *embeddings,
                                              const int32_t *offsets, float *out, float *inputs01, const int32_t *B_dev,
                                              uint32_t B_cap, uint32_t stride, uint32_t L, uint32_t max_level, float S,
                                              uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                              void *binned_workspace, uint32_t n_rows_total, ngp_stream_t stream)
{
    return ngp_x_grid_encode_forward_slab_jac(xyzs, bound, embeddings, offsets, out, inputs01, B_dev, B_cap, stride, L,
                                              max_level, S, H, gridtype, align_corners, interp, binned_workspace,
                                              n_rows_total, nullptr, stream);
}

// Levels level_lo .. level_hi - 1 only (the others' slab rows stay as they are): the data-parallel step encodes the levels
// whose parameters have arrived while the all-gather of the others is still on the wire.  x01 is written with level 0.
extern "C" int ngp_x_grid_encode_forward_slab_levels(const float *xyzs, float bound, const float *embeddings,
                                                     const int32_t *offsets, float *out, float *inputs01,
                                                     const int32_t *B_dev, uint32_t B_cap, uint32_t stride, uint32_t L,
                                                     uint32_t level_lo, uint32_t level_hi, float S, uint32_t H,
                                                     uint32_t gridtype, int align_corners, uint32_t interp, float *dydx,
                                                     ngp_stream_t stream)
{
    if (B_cap == 0 || level_hi <= level_lo) return NGP_OK;
    NGP_REQUIRE(xyzs && embeddings && offsets && out, "grid_encode_forward_slab_levels: null tensor");
    NGP_REQUIRE(stride >= B_cap, "grid_encode_forward_slab_levels: stride smaller than B_cap");
    NGP_REQUIRE(bound > 0.0f, "grid_encode_forward_slab_levels: bound must be positive");
    LevelRes lv;
    NGP_REQUIRE(fill_levels(lv, S, H, L), "grid_encode_forward_slab_levels: L must be in [1, %u]", kMaxLevels);
    NGP_REQUIRE(level_hi <= L, "grid_encode_forward_slab_levels: level_hi > L");
    const uint32_t nchunks = ceil_div(B_cap, kBlock), n = level_hi - level_lo;
    static const bool snake_on = !(getenv("NGP_SNAKE") && getenv("NGP_SNAKE")[0] == '0');
    const uint32_t snake_levels = (snake_on && n >= 8) ? n : 0u;
    const uint32_t pblocks = slab_persistent_blocks(B_dev != nullptr && snake_levels != 0);
    const dim3 grid(pblocks ? pblocks : (snake_levels ? snake_blocks(n, nchunks) : nchunks * n));
    if (dydx)
        grid_forward_slab_kernel<false, true><<<grid, dim3(kBlock), 0, as_stream(stream)>>>(
            xyzs, bound, embeddings, offsets, out, inputs01, B_dev, B_cap, stride, nchunks, lv, gridtype, align_corners != 0,
            interp, WsLayout{}, dydx, snake_levels, 0u, LevelPlacement{}, level_lo, pblocks ? 1u : 0u);
    else
        grid_forward_slab_kernel<false><<<grid, dim3(kBlock), 0, as_stream(stream)>>>(
            xyzs, bound, embeddings, offsets, out, inputs01, B_dev, B_cap, stride, nchunks, lv, gridtype, align_corners != 0,
            interp, WsLayout{}, nullptr, snake_levels, 0u, LevelPlacement{}, level_lo, pblocks ? 1u : 0u);
    NGP_CHECK_LAUNCH("grid_encode_forward_slab_levels");
    return NGP_OK;
}

extern "C" int ngp_x_grid_encode_forward_slab_jac(const float *xyzs, float bound, const float *embeddings,
                                                  const int32_t *offsets, float *out, float *inputs01,
                                                  const int32_t *B_dev, uint32_t B_cap, uint32_t stride, uint32_t L,
                                                  uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                                  int align_corners, uint32_t interp, void *binned_workspace,
                                                  uint32_t n_rows_total, float *dydx, ngp_stream_t stream)
{
    return ngp_x_grid_encode_forward_slab_placed(xyzs, bound, embeddings, offsets, out, inputs01, B_dev, B_cap, stride, L,
                                                 max_level, S, H, gridtype, align_corners, interp, binned_workspace,
                                                 n_rows_total, dydx, nullptr, stream);
}

extern "C" int ngp_x_grid_encode_forward_slab_placed(const float *xyzs, float bound, const float *embeddings,
                                                     const int32_t *offsets, float *out, float *inputs01,
                                                     const int32_t *B_dev, uint32_t B_cap, uint32_t stride, uint32_t L,
                                                     uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                                     int align_corners, uint32_t interp, void *binned_workspace,
                                                     uint32_t n_rows_total, float *dydx, const float *level_cost,
                                                     ngp_stream_t stream)
{
    if (B_cap == 0 || max_level == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && embeddings && offsets && out, "grid_encode_forward_slab: null tensor");
    NGP_REQUIRE(stride >= B_cap, "grid_encode_forward_slab: stride smaller than B_cap");
    NGP_REQUIRE(bound > 0.0f, "grid_encode_forward_slab: bound must be positive");
    LevelRes lv;
    NGP_REQUIRE(fill_levels(lv, S, H, L), "grid_encode_forward_slab: L must be in [1, %u]", kMaxLevels);
    NGP_REQUIRE(max_level <= L, "grid_encode_forward_slab: max_level > L");
    const uint32_t nchunks = ceil_div(B_cap, kBlock);
    // level -> XCD placement: snake (XCD k takes levels k and 15 - k: a cheap coarse level paired with an expensive fine one)
    // instead of the contiguous split (levels 2k, 2k + 1: XCD 7 gets the two most expensive levels and decides when the
    // kernel ends) -- 61 -> 46 us in the step; NGP_SNAKE=0 restores the contiguous split (fewer than 8 levels always take it)
    static const bool snake_on = !(getenv("NGP_SNAKE") && getenv("NGP_SNAKE")[0] == '0');
    const uint32_t snake_levels = (snake_on && max_level >= 8) ? max_level : 0u;
    dim3 grid(snake_levels ? snake_blocks(max_level, nchunks) : nchunks * max_level);
    // level_cost (host, max_level floats > 0): the caller's measured cost of one tile of each level for ITS points -- the
    // levels are then dealt to the XCDs in runs of equal cost (ngp_common.hpp: place_levels) instead of by the snake
    LevelPlacement place = {};
    uint32_t placed_levels = 0;
    if (level_cost && max_level >= 8 && max_level <= kPlacedLevels) {
        for (uint32_t l = 0; l < max_level; l++)
            NGP_REQUIRE(level_cost[l] > 0.0f && level_cost[l] < 1e30f, "grid_encode_forward_slab: level_cost[%u] must be positive", l);
        placed_levels = max_level;
        grid = dim3(place_levels(place, level_cost, max_level, nchunks));
        if (const char *only = getenv("NGP_PLACE_ONLY_LEVEL")) {   // diagnostic (tools/level_costs.py): ONE level, on XCD 0
            const uint32_t l = (uint32_t)atoi(only);
            NGP_REQUIRE(l < max_level, "NGP_PLACE_ONLY_LEVEL out of range");
            place = LevelPlacement{};
            place.own[0][l] = 0xff;
            grid = dim3(8u * nchunks);
        }
        if (getenv("NGP_PLACE_SPREAD")) {   // diagnostic: every XCD works on an eighth of every level, level by level
            place = LevelPlacement{};
            for (uint32_t k = 0; k < 8; k++)
                for (uint32_t l = 0; l < max_level; l++) place.own[k][l] = (uint8_t)(1u << k);
            grid = dim3(8u * max_level * ceil_div(nchunks, 8u));
        }
    }
    // a device-side point count under the snake: persistent workgroups (see the kernel)
    const dim3 pgrid(slab_persistent_blocks(B_dev != nullptr && snake_levels != 0 && placed_levels == 0 && !binned_workspace));
    if (binned_workspace) {
        // the workspace of ngp_x_grid_backward_binned_* for the same samples, planned (mode 2 of prepare): count here
        NGP_REQUIRE(max_level == L && ((uintptr_t)binned_workspace & 15u) == 0 && n_rows_total > 0,
                    "grid_encode_forward_slab: counting needs max_level == L and an aligned binned workspace");
        const uint32_t n_chunks_max = n_rows_total / kChunkRows + L + 1;
        NGP_REQUIRE(n_chunks_max <= kMaxChunks, "grid_encode_forward_slab: table too large for the binned backward");
        const WsLayout w = ws_layout(binned_workspace, n_chunks_max);
        if (dydx)
            grid_forward_slab_kernel<true, true><<<grid, dim3(kBlock), n_chunks_max * 4, as_stream(stream)>>>(
                xyzs, bound, embeddings, offsets, out, inputs01, B_dev, B_cap, stride, nchunks, lv, gridtype,
                align_corners != 0, interp, w, dydx, snake_levels, placed_levels, place);
        else
            grid_forward_slab_kernel<true><<<grid, dim3(kBlock), n_chunks_max * 4, as_stream(stream)>>>(
                xyzs, bound, embeddings, offsets, out, inputs01, B_dev, B_cap, stride, nchunks, lv, gridtype,
                align_corners != 0, interp, w, nullptr, snake_levels, placed_levels, place);
    } else if (dydx) {
        grid_forward_slab_kernel<false, true><<<pgrid.x ? pgrid : grid, dim3(kBlock), 0, as_stream(stream)>>>(
            xyzs, bound, embeddings, offsets, out, inputs01, B_dev, B_cap, stride, nchunks, lv, gridtype,
            align_corners != 0, interp, WsLayout{}, dydx, snake_levels, placed_levels, place, 0u, pgrid.x ? 1u : 0u);
    } else {
        grid_forward_slab_kernel<false><<<pgrid.x ? pgrid : grid, dim3(kBlock), 0, as_stream(stream)>>>(
            xyzs, bound, embeddings, offsets, out, inputs01, B_dev, B_cap, stride, nchunks, lv, gridtype,
            align_corners != 0, interp, WsLayout{}, nullptr, snake_levels, placed_levels, place, 0u, pgrid.x ? 1u : 0u);
    }
    NGP_CHECK_LAUNCH("grid_encode_forward_slab");
    return NGP_OK;
}

extern "C" int ngp_x_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *ts,
                                                  const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                                  float *weights, float *weights_sum, float *depth, float *image,
                                                  ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays && weights_sum && depth && image, "x_composite_rays_train_forward: null tensor");
    NGP_REQUIRE(M == 0 || (sigmas && rgbs && ts && weights), "x_composite_rays_train_forward: null sample tensor");
    composite_forward_wave_kernel<<<dim3(ceil_div(N, 4u)), dim3(256), 0, as_stream(stream)>>>(
        sigmas, rgbs, ts, rays, M, N, T_thresh, weights, weights_sum, depth, image);
    NGP_CHECK_LAUNCH("x_composite_rays_train_forward");
    return NGP_OK;
}

extern "C" int ngp_x_composite_rays_train_backward(const float *grad_weights, const float *grad_weights_sum,
                                                   const float *grad_depth, const float *grad_image,
                                                   const float *sigmas, const float *rgbs, const float *ts,
                                                   const int32_t *rays, const float *weights_sum, const float *depth,
                                                   const float *image, uint32_t M, uint32_t N, float T_thresh,
                                                   float *grad_sigmas, float *grad_rgbs, ngp_stream_t stream)
{
    if (N == 0 || M == 0) return NGP_OK;
    NGP_REQUIRE(grad_weights && grad_weights_sum && grad_depth && grad_image && sigmas && rgbs && ts && rays &&
                    weights_sum && depth && image && grad_sigmas && grad_rgbs,
                "x_composite_rays_train_backward: null tensor");
    composite_backward_wave_kernel<0><<<dim3(ceil_div(N, kCompBwdBlock / 64)), dim3(kCompBwdBlock), 0, as_stream(stream)>>>(
        grad_weights, grad_weights_sum, grad_depth, grad_image, nullptr, nullptr, 0.0f, sigmas, rgbs, ts, rays,
        weights_sum, depth, image, M, N, T_thresh, grad_sigmas, grad_rgbs, nullptr, nullptr, nullptr, nullptr);
    NGP_CHECK_LAUNCH("x_composite_rays_train_backward");
    return NGP_OK;
}

extern "C" int ngp_x_composite_mse_backward(const float *gt_rgba, const float *bg_rgb, float bg_const,
                                            const float *sigmas, const float *rgbs, const float *ts,
                                            const int32_t *rays, const float *weights_sum, const float *depth,
                                            const float *image, uint32_t M, uint32_t N, float T_thresh,
                                            float *grad_sigmas, float *grad_rgbs, float *loss_out, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(gt_rgba && rays && weights_sum && depth && image && loss_out, "composite_mse_backward: null tensor");
    NGP_REQUIRE(M == 0 || (sigmas && rgbs && ts && grad_sigmas && grad_rgbs), "composite_mse_backward: null sample tensor");
    NGP_REQUIRE(((uintptr_t)gt_rgba & 15u) == 0, "composite_mse_backward: gt_rgba must be 16-byte aligned");
    composite_backward_wave_kernel<1><<<dim3(ceil_div(N, kCompBwdBlock / 64)), dim3(kCompBwdBlock), 0, as_stream(stream)>>>(
        nullptr, nullptr, nullptr, nullptr, gt_rgba, bg_rgb, bg_const, sigmas, rgbs, ts, rays, weights_sum, depth, image, M,
        N, T_thresh, grad_sigmas, grad_rgbs, loss_out, nullptr, nullptr, nullptr);
    NGP_CHECK_LAUNCH("composite_mse_backward");
    return NGP_OK;
}

extern "C" int ngp_x_composite_mse_train(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *sigmas,
                                         const float *rgbs, const float *ts, const int32_t *rays, uint32_t M, uint32_t N,
                                         float T_thresh, float *weights_sum, float *depth, float *image,
                                         float *grad_sigmas, float *grad_rgbs, float *loss_out, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(gt_rgba && rays && weights_sum && depth && image && loss_out, "composite_mse_train: null tensor");
    NGP_REQUIRE(M == 0 || (sigmas && rgbs && ts && grad_sigmas && grad_rgbs), "composite_mse_train: null sample tensor");
    NGP_REQUIRE(((uintptr_t)gt_rgba & 15u) == 0, "composite_mse_train: gt_rgba must be 16-byte aligned");
    composite_backward_wave_kernel<2><<<dim3(ceil_div(N, kCompBwdBlock / 64)), dim3(kCompBwdBlock), 0, as_stream(stream)>>>(
        nullptr, nullptr, nullptr, nullptr, gt_rgba, bg_rgb, bg_const, sigmas, rgbs, ts, rays, nullptr, nullptr, nullptr, M,
        N, T_thresh, grad_sigmas, grad_rgbs, loss_out, weights_sum, depth, image);
    NGP_CHECK_LAUNCH("composite_mse_train");
    return NGP_OK;
}

// ngp_x_composite_mse_train + the list of samples in front of the early stop (live_n [N], live_idx [M], live_count [1])
extern "C" int ngp_x_composite_mse_train_idx(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *sigmas,
                                             const float *rgbs, const float *ts, const int32_t *rays, uint32_t M, uint32_t N,
                                             float T_thresh, float *weights_sum, float *depth, float *image,
                                             float *grad_sigmas, float *grad_rgbs, float *loss_out, int32_t *live_n,
                                             int32_t *live_idx, int32_t *live_count, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(gt_rgba && rays && weights_sum && depth && image && loss_out && live_n && live_idx && live_count,
                "composite_mse_train_idx: null tensor");
    NGP_REQUIRE(M > 0 && sigmas && rgbs && ts && grad_sigmas && grad_rgbs, "composite_mse_train_idx: null sample tensor");
    NGP_REQUIRE(((uintptr_t)gt_rgba & 15u) == 0, "composite_mse_train_idx: gt_rgba must be 16-byte aligned");
    HdrLoss hdr;
    hdr.live_out = live_n;
    composite_backward_wave_kernel<2><<<dim3(ceil_div(N, kCompBwdBlock / 64)), dim3(kCompBwdBlock), 0, as_stream(stream)>>>(
        nullptr, nullptr, nullptr, nullptr, gt_rgba, bg_rgb, bg_const, sigmas, rgbs, ts, rays, nullptr, nullptr, nullptr, M,
        N, T_thresh, grad_sigmas, grad_rgbs, loss_out, weights_sum, depth, image, hdr);
    live_index_kernel<<<dim3(ceil_div(N, 16u)), dim3(1024), 0, as_stream(stream)>>>(rays, live_n, N, M, live_idx, live_count,
                                                                                    nullptr);
    NGP_CHECK_LAUNCH("composite_mse_train_idx");
    return NGP_OK;
}

extern "C" int ngp_x_composite_hdr_train(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *exposure,
                                         const float *weight, float inv_norm, const float *sigmas, const float *rgbs,
                                         const float *ts, const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                         float *weights_sum, float *depth, float *image, float *grad_sigmas,
                                         float *grad_rgbs, float *loss_out, ngp_stream_t stream)
{
    NGP_REQUIRE(exposure, "composite_hdr_train: null tensor");
    return ngp_x_composite_train_live(gt_rgba, bg_rgb, bg_const, exposure, weight, inv_norm, nullptr, 0.0f, sigmas, rgbs, ts,
                                      rays, M, N, T_thresh, weights_sum, depth, image, grad_sigmas, grad_rgbs, loss_out, stream);
}

extern "C" int ngp_x_composite_train_live(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *exposure,
                                          const float *weight, float inv_norm, const int32_t *n_live, float lambda_entropy,
                                          const float *sigmas, const float *rgbs, const float *ts, const int32_t *rays,
                                          uint32_t M, uint32_t N, float T_thresh, float *weights_sum, float *depth,
                                          float *image, float *grad_sigmas, float *grad_rgbs, float *loss_out,
                                          ngp_stream_t stream)
{
    return ngp_x_composite_train_live_idx(gt_rgba, bg_rgb, bg_const, exposure, weight, inv_norm, n_live, lambda_entropy, sigmas,
                                          rgbs, ts, rays, M, N, T_thresh, weights_sum, depth, image, grad_sigmas, grad_rgbs,
                                          loss_out, nullptr, nullptr, nullptr, nullptr, stream);
}

// ... that also lists the samples in front of the early stop (as ngp_x_composite_mse_train_idx; live_off [N]: where each
// ray's entries start in the list, for per-ray consumers such as ngp_x_ray_gradients_list).  live_n NULL: no list.
extern "C" int ngp_x_composite_train_live_idx(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *exposure,
                                              const float *weight, float inv_norm, const int32_t *n_live,
                                              float lambda_entropy, const float *sigmas, const float *rgbs, const float *ts,
                                              const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                              float *weights_sum, float *depth, float *image, float *grad_sigmas,
                                              float *grad_rgbs, float *loss_out, int32_t *live_n, int32_t *live_idx,
                                              int32_t *live_count, int32_t *live_off, ngp_stream_t stream)
{
    return ngp_x_composite_train_terms(gt_rgba, bg_rgb, bg_const, exposure, weight, inv_norm, n_live, lambda_entropy, nullptr,
                                       0.0f, sigmas, rgbs, ts, rays, M, N, T_thresh, weights_sum, depth, image, grad_sigmas,
                                       grad_rgbs, loss_out, live_n, live_idx, live_count, live_off, nullptr, stream);
}

// ... plus a term over the samples' compositing weights: loss += lambda_sample * sum_i weights[i] * sample_term[i]
// (sample_term [M], e.g. ngp_x_orientation_term; NULL: none)
extern "C" int ngp_x_composite_train_terms(const float *gt_rgba, const float *bg_rgb, float bg_const, const float *exposure,
                                           const float *weight, float inv_norm, const int32_t *n_live, float lambda_entropy,
                                           const float *sample_term, float lambda_sample, const float *sigmas,
                                           const float *rgbs, const float *ts, const int32_t *rays, uint32_t M, uint32_t N,
                                           float T_thresh, float *weights_sum, float *depth, float *image,
                                           float *grad_sigmas, float *grad_rgbs, float *loss_out, int32_t *live_n,
                                           int32_t *live_idx, int32_t *live_count, int32_t *live_off, float *term_weight,
                                           ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(gt_rgba && rays && weights_sum && depth && image && loss_out, "composite_train_live: null tensor");
    NGP_REQUIRE(M == 0 || (sigmas && rgbs && ts && grad_sigmas && grad_rgbs), "composite_train_live: null sample tensor");
    NGP_REQUIRE(((uintptr_t)gt_rgba & 15u) == 0, "composite_train_live: gt_rgba must be 16-byte aligned");
    NGP_REQUIRE(!exposure || n_live || inv_norm > 0.0f, "composite_train_live: inv_norm = 1 / sum(lossmult) must be positive");
    NGP_REQUIRE(!(weight && n_live), "composite_train_live: a Bayer / loss weight and adaptive batches are not combined");
    HdrLoss hdr;
    hdr.exposure = exposure;
    hdr.weight = weight;
    hdr.inv_norm = inv_norm;
    hdr.n_live = n_live;
    NGP_REQUIRE(lambda_entropy >= 0.0f, "composite_train_live: lambda_entropy must not be negative");
    hdr.lambda_entropy = lambda_entropy;
    NGP_REQUIRE(sample_term ? lambda_sample >= 0.0f : lambda_sample == 0.0f,
                "composite_train_terms: lambda_sample needs sample_term and must not be negative");
    NGP_REQUIRE(!term_weight || sample_term, "composite_train_terms: term_weight needs sample_term");
    hdr.sample_term = sample_term;
    hdr.lambda_sample = lambda_sample;
    hdr.term_weight_out = term_weight;
    const bool listing = live_n != nullptr;
    NGP_REQUIRE(listing ? (live_idx && live_count && M > 0) : (!live_idx && !live_count && !live_off),
                "composite_train_live: live_n, live_idx and live_count go together");
    hdr.live_out = live_n;
    composite_backward_wave_kernel<2><<<dim3(ceil_div(N, kCompBwdBlock / 64)), dim3(kCompBwdBlock), 0, as_stream(stream)>>>(
        nullptr, nullptr, nullptr, nullptr, gt_rgba, bg_rgb, bg_const, sigmas, rgbs, ts, rays, nullptr, nullptr, nullptr, M,
        N, T_thresh, grad_sigmas, grad_rgbs, loss_out, weights_sum, depth, image, hdr);
    if (listing)
        live_index_kernel<<<dim3(ceil_div(N, 16u)), dim3(1024), 0, as_stream(stream)>>>(rays, live_n, N, M, live_idx, live_count,
                                                                                        live_off);
    NGP_CHECK_LAUNCH("composite_train_live");
    return NGP_OK;
}

extern "C" int ngp_x_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, uint64_t n, float lr,
                               double beta1, double beta2, float eps, uint32_t step, int zero_grad, ngp_stream_t stream)
{
    if (n == 0) return NGP_OK;
    NGP_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam_step: null tensor");
    NGP_REQUIRE(step >= 1, "adam_step: step counts from 1");
    NGP_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u) == 0,
                "adam_step: tensors must be 16-byte aligned");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const size_t n4 = n / 4;
    const uint32_t blocks = (uint32_t)min((size_t)256 * 8, (n4 + 255) / 256 + 1);
    adam_kernel<<<dim3(blocks), dim3(256), 0, as_stream(stream)>>>(param, grad, exp_avg, exp_avg_sq, n4, n, lr, (float)beta1,
                                                                  (float)beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)), zero_grad != 0,
                                                                  grad, nullptr, nullptr);
    NGP_CHECK_LAUNCH("adam_step");
    return NGP_OK;
}

extern "C" int ngp_x_adam_step_dev(float *param, float *grad, float *exp_avg, float *exp_avg_sq, uint64_t n,
                                   const float *hyper, float beta1, float beta2, float eps, int zero_grad,
                                   const uint32_t *skip, ngp_stream_t stream)
{
    if (n == 0) return NGP_OK;
    NGP_REQUIRE(param && grad && exp_avg && exp_avg_sq && hyper, "adam_step_dev: null tensor");
    NGP_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u) == 0,
                "adam_step_dev: tensors must be 16-byte aligned");
    const size_t n4 = n / 4;
    const uint32_t blocks = (uint32_t)min((size_t)256 * 8, (n4 + 255) / 256 + 1);
    adam_kernel<<<dim3(blocks), dim3(256), 0, as_stream(stream)>>>(param, grad, exp_avg, exp_avg_sq, n4, n, 0.0f, beta1,
                                                                  beta2, eps, 1.0f, 1.0f, zero_grad != 0, grad, hyper, skip);
    NGP_CHECK_LAUNCH("adam_step_dev");
    return NGP_OK;
}

extern "C" int ngp_x_schedule_step(uint32_t *step_counter, float *hyper, double lr0, double decay_steps, double beta1,
                                   double beta2, ngp_stream_t stream)
{
    NGP_REQUIRE(step_counter && hyper, "schedule_step: null tensor");
    NGP_REQUIRE(decay_steps > 0.0, "schedule_step: decay_steps must be positive");
    schedule_kernel<<<dim3(1), dim3(1), 0, as_stream(stream)>>>(step_counter, hyper, lr0, decay_steps, beta1, beta2);
    NGP_CHECK_LAUNCH("schedule_step");
    return NGP_OK;
}

int ngp::step_begin_args(StepBegin &a, const char *who, uint32_t *step_counter, float *hyper, double lr0, double decay_steps,
                         double beta1, double beta2, float *loss_out, int64_t *samples_seen, const int32_t *sample_counter,
                         void *binned_workspace, uint32_t L, uint32_t n_rows_total, int single_segment, float *scaler,
                         double growth, double backoff, uint32_t growth_interval)
{
    NGP_REQUIRE(step_counter && hyper, "%s: null tensor", who);
    NGP_REQUIRE(!scaler || (growth >= 1.0 && backoff > 0.0 && backoff <= 1.0 && growth_interval >= 1),
                "%s: loss scaler: growth >= 1, 0 < backoff <= 1, growth_interval >= 1", who);
    NGP_REQUIRE(decay_steps > 0.0, "%s: decay_steps must be positive", who);
    NGP_REQUIRE((samples_seen == nullptr) == (sample_counter == nullptr), "%s: samples_seen and sample_counter go together", who);
    WsLayout w{};
    if (binned_workspace) {   // = ngp_x_grid_backward_binned_prepare(stage 2) for that workspace
        NGP_REQUIRE(L >= 1 && L <= kMaxLevels && n_rows_total > 0 && ((uintptr_t)binned_workspace & 15u) == 0,
                    "%s: bad binned workspace arguments", who);
        const uint32_t n_chunks_max = n_rows_total / kChunkRows + L + 1;
        NGP_REQUIRE(n_chunks_max <= kMaxChunks, "%s: table too large for the binned backward", who);
        w = ws_layout(binned_workspace, n_chunks_max);
    }
    a = StepBegin{step_counter, hyper, lr0, decay_steps, beta1, beta2, loss_out, reinterpret_cast<long long *>(samples_seen),
                  sample_counter, binned_workspace != nullptr, single_segment != 0, L, w, scaler, (float)growth, (float)backoff,
                  growth_interval};
    return NGP_OK;
}

extern "C" int ngp_x_step_begin(uint32_t *step_counter, float *hyper, double lr0, double decay_steps, double beta1,
                                double beta2, float *loss_out, int64_t *samples_seen, const int32_t *sample_counter,
                                void *binned_workspace, uint32_t L, uint32_t n_rows_total, int single_segment,
                                float *scaler, double growth, double backoff, uint32_t growth_interval, ngp_stream_t stream)
{
    StepBegin a;
    const int rc = step_begin_args(a, "step_begin", step_counter, hyper, lr0, decay_steps, beta1, beta2, loss_out, samples_seen,
                                   sample_counter, binned_workspace, L, n_rows_total, single_segment, scaler, growth, backoff,
                                   growth_interval);
    if (rc != NGP_OK) return rc;
    step_begin_kernel<<<dim3(1), dim3(binned_workspace ? 1024 : 64), 0, as_stream(stream)>>>(a);
    NGP_CHECK_LAUNCH("step_begin");
    return NGP_OK;
}

extern "C" int ngp_x_adam_step_dev2(float *param_a, float *grad_a, float *exp_avg_a, float *exp_avg_sq_a, uint64_t n_a,
                                    int zero_grad_a, float *param_b, float *grad_b, float *exp_avg_b,
                                    float *exp_avg_sq_b, uint64_t n_b, int zero_grad_b, const float *hyper, float beta1,
                                    float beta2, float eps, int grad_a_bf16, const uint32_t *skip, ngp_stream_t stream)
{
    NGP_REQUIRE(n_a > 0 && n_b > 0, "adam_step_dev2: empty tensor (use adam_step_dev)");
    NGP_REQUIRE(!(grad_a_bf16 && zero_grad_a), "adam_step_dev2: a bfloat16 gradient is not zeroed");
    NGP_REQUIRE(param_a && grad_a && exp_avg_a && exp_avg_sq_a && param_b && grad_b && exp_avg_b && exp_avg_sq_b && hyper,
                "adam_step_dev2: null tensor");
    NGP_REQUIRE((((uintptr_t)param_a | (uintptr_t)grad_a | (uintptr_t)exp_avg_a | (uintptr_t)exp_avg_sq_a |
                  (uintptr_t)param_b | (uintptr_t)grad_b | (uintptr_t)exp_avg_b | (uintptr_t)exp_avg_sq_b) & 15u) == 0,
                "adam_step_dev2: tensors must be 16-byte aligned");
    const auto blocks_for = [](uint64_t n) { return (uint32_t)min((size_t)256 * 8, (size_t)(n / 4 + 255) / 256 + 1); };
    const uint32_t ba = blocks_for(n_a), bb = blocks_for(n_b);
    const AdamTensor a{param_a, grad_a, exp_avg_a, exp_avg_sq_a, (size_t)n_a, zero_grad_a != 0};
    const AdamTensor b{param_b, grad_b, exp_avg_b, exp_avg_sq_b, (size_t)n_b, zero_grad_b != 0};
    if (grad_a_bf16)
        adam2_kernel<true><<<dim3(ba + bb), dim3(256), 0, as_stream(stream)>>>(a, b, ba, beta1, beta2, eps, hyper, skip);
    else
        adam2_kernel<false><<<dim3(ba + bb), dim3(256), 0, as_stream(stream)>>>(a, b, ba, beta1, beta2, eps, hyper, skip);
    NGP_CHECK_LAUNCH("adam_step_dev2");
    return NGP_OK;
}

extern "C" int ngp_x_counter_add(uint32_t *counter, uint32_t delta, ngp_stream_t stream)
{
    NGP_REQUIRE(counter, "counter_add: null tensor");
    counter_add_kernel<<<dim3(1), dim3(1), 0, as_stream(stream)>>>(counter, delta);
    NGP_CHECK_LAUNCH("counter_add");
    return NGP_OK;
}

extern "C" int ngp_x_sample_rays(const uint8_t *images, uint32_t V, uint32_t H, uint32_t W, uint32_t C, const float *poses,
                                 float fx, float fy, float cx, float cy, uint32_t N, uint64_t seed,
                                 const uint32_t *draw_dev, uint32_t draw, float *rays_o, float *rays_d, float *gt_rgba,
                                 float *noises, float *bg_rgb, int32_t *index, ngp_stream_t stream)
{
    return ngp_x_sample_rays_lit(images, V, H, W, C, poses, fx, fy, cx, cy, N, seed, draw_dev, draw, rays_o, rays_d, gt_rgba,
                                 noises, bg_rgb, index, nullptr, nullptr, stream);
}

extern "C" int ngp_x_sample_rays_lit(const uint8_t *images, uint32_t V, uint32_t H, uint32_t W, uint32_t C,
                                     const float *poses, float fx, float fy, float cx, float cy, uint32_t N, uint64_t seed,
                                     const uint32_t *draw_dev, uint32_t draw, float *rays_o, float *rays_d, float *gt_rgba,
                                     float *noises, float *bg_rgb, int32_t *index, const float *view_ldirs,
                                     float *rays_ldir, ngp_stream_t stream)
{
    return ngp_x_sample_rays_adaptive(images, V, H, W, C, poses, fx, fy, cx, cy, N, seed, draw_dev, draw, rays_o, rays_d,
                                      gt_rgba, noises, bg_rgb, index, view_ldirs, rays_ldir, nullptr, nullptr, nullptr, 0,
                                      nullptr, nullptr, stream);
}

extern "C" int ngp_x_sample_rays_adaptive(const uint8_t *images, uint32_t V, uint32_t H, uint32_t W, uint32_t C,
                                          const float *poses, float fx, float fy, float cx, float cy, uint32_t N,
                                          uint64_t seed, const uint32_t *draw_dev, uint32_t draw, float *rays_o,
                                          float *rays_d, float *gt_rgba, float *noises, float *bg_rgb, int32_t *index,
                                          const float *view_ldirs, float *rays_ldir, const int32_t *prev_samples,
                                          const int32_t *prev_live, int32_t *live, uint32_t num_points,
                                          const float *view_exposure, float *exposure, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE((view_exposure == nullptr) == (exposure == nullptr), "sample_rays: view_exposure and exposure go together");
    NGP_REQUIRE((prev_samples == nullptr) == (prev_live == nullptr), "sample_rays: prev_samples and prev_live go together");
    NGP_REQUIRE(!prev_samples || (live && num_points > 0), "sample_rays: adaptive batches need `live` and num_points");
    AdaptiveRays ad;
    ad.prev_samples = prev_samples;
    ad.prev_live = prev_live;
    ad.live = live;
    ad.num_points = num_points;
    NGP_REQUIRE((view_ldirs == nullptr) == (rays_ldir == nullptr), "sample_rays: view_ldirs and rays_ldir go together");
    NGP_REQUIRE(images && poses && rays_o && rays_d && gt_rgba, "sample_rays: null tensor");
    NGP_REQUIRE(V > 0 && H > 0 && W > 0 && (uint64_t)H * W < (1ull << 32), "sample_rays: bad image shape");
    NGP_REQUIRE(C == 3 || C == 4, "sample_rays: images must be RGB or RGBA (uint8)");
    if (prev_live && prev_live == live) {   // one ray slot: the count is one device word, see adaptive_live_kernel
        adaptive_live_kernel<<<dim3(1), dim3(1), 0, as_stream(stream)>>>(ad, N);
        NGP_CHECK_LAUNCH("sample_rays (live count)");
        ad.resolved = true;
    }
    sample_rays_kernel<<<dim3(ceil_div(N, 256u)), dim3(256), 0, as_stream(stream)>>>(
        images, V, H, W, C, poses, fx, fy, cx, cy, N, (uint32_t)seed, (uint32_t)(seed >> 32), draw_dev, draw, rays_o, rays_d,
        gt_rgba, noises, bg_rgb, index, view_ldirs, rays_ldir, ad, view_exposure, exposure);
    NGP_CHECK_LAUNCH("sample_rays");
    return NGP_OK;
}

extern "C" int ngp_x_near_far_from_aabb_v2(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                                           float min_near, float *nears, float *fars, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && aabb && nears && fars, "near_far_from_aabb_v2: null tensor");
    near_far_v2_kernel<<<dim3(ceil_div(N, 256u)), dim3(256), 0, as_stream(stream)>>>(rays_o, rays_d, aabb, N, min_near,
                                                                                    nears, fars);
    NGP_CHECK_LAUNCH("near_far_from_aabb_v2");
    return NGP_OK;
}
