// Multiresolution hash / tiled grid encoder for gfx950.
//
// Behaviour follows the reference kernels (file:line into the reference tree):
//   forward + dy_dx     gridencoder/src/gridencoder.cu:82-249
//   table gradient      :252-349        input gradient :352-378
//   total variation     :525-631        weight decay   :670-703
//   hash / index rule   :45-79
// The design does not: one 1-D grid whose (level, chunk) work items are dealt to the
// 8 XCDs as contiguous level ranges (each XCD keeps the one or two 4 MiB level tables it
// is working on in its own L2), per-level geometry (row count, strides, hash / mask /
// modulo mode) derived once per workgroup in scalar registers, whole-row vector gathers
// (float2 for C = 2) all issued before the first use, streaming (non-temporal) stores.
// Arithmetic choices (explicit fmaf, no implicit contraction) mirror oracle/ngp_oracle.c.
#include "grid_common.hpp"

namespace ngp {

// ------------------------------------------------------------------ forward
// LM: the Jacobian is stored level-major, dy_dx[level, b, d, ch] (a private layout between the two x_ entry points: the
// lanes of a wave then write one contiguous run instead of 6 floats every 384 bytes)
template <uint32_t D, uint32_t C, bool JAC, bool LM = false>
__global__ __launch_bounds__(kBlock) void grid_forward_kernel(
    const float *__restrict__ inputs, const float *__restrict__ table, const int32_t *__restrict__ offsets,
    float *__restrict__ outputs, float *__restrict__ dy_dx, uint32_t B, uint32_t L, uint32_t nchunks,
    uint32_t max_level, LevelRes lv, uint32_t gridtype, bool align_corners, uint32_t interp)
{
    uint32_t level, chunk;
    snake_level_tile(blockIdx.x, nchunks, max_level, level, chunk);
    if (level == kNoLevel) return;
    const uint32_t b = chunk * kBlock + threadIdx.x;
    if (b >= B) return;

    const Geom<D> g = make_geom<D>(offsets, level, lv.res[level], gridtype);
    const float *__restrict__ tab = table + (size_t)(uint32_t)offsets[level] * C;

    float x[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];

    float *out = outputs + ((size_t)level * B + b) * C;
    float *jac = !JAC ? nullptr
                 : LM ? dy_dx + ((size_t)level * B + b) * D * C
                      : dy_dx + (size_t)b * L * D * C + (size_t)level * D * C;

    Cell<D> cl;
    if (!locate<D>(x, g.res, align_corners, interp, cl)) {
        Row<C> z;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) z.v[ch] = 0.0f;
        z.store_stream(out);
        if (JAC) {
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) jac[i] = 0.0f;
        }
        return;
    }

    // all 2^D row gathers are issued before the first multiply
    constexpr uint32_t NCORN = 1u << D;
    Row<C> rows[NCORN];
    float wts[NCORN];
#pragma unroll
    for (uint32_t corner = 0; corner < NCORN; corner++) {
        float w = 1.0f;
        uint32_t c[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if (corner & (1u << d)) {
                w *= cl.f[d];
                c[d] = min(cl.c[d] + 1u, g.res - 1u);
            } else {
                w *= 1.0f - cl.f[d];
                c[d] = cl.c[d];
            }
        }
        wts[corner] = w;
        rows[corner].load(tab + (size_t)row_of<D>(g, c) * C);
    }
    Row<C> acc;
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) acc.v[ch] = 0.0f;
#pragma unroll
    for (uint32_t corner = 0; corner < NCORN; corner++)
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) acc.v[ch] = fmaf(wts[corner], rows[corner].v[ch], acc.v[ch]);
    acc.store_stream(out);

    if constexpr (JAC) {
        // d out / d x_gd: the corner rows are already in registers (corner bit gd = 0 / 1)
        const float scale = (float)(align_corners ? g.res - 1u : g.res);
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            float gacc[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) gacc[ch] = 0.0f;
#pragma unroll
            for (uint32_t combo = 0; combo < (1u << (D - 1)); combo++) {
                float w = scale;
                uint32_t lo = 0;
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = nd >= gd ? nd + 1 : nd;
                    if (combo & (1u << nd)) {
                        w *= cl.f[d];
                        lo |= 1u << d;
                    } else {
                        w *= 1.0f - cl.f[d];
                    }
                }
                const uint32_t hi = lo | (1u << gd);
#pragma unroll
                for (uint32_t ch = 0; ch < C; ch++)
                    gacc[ch] = fmaf(w * (rows[hi].v[ch] - rows[lo].v[ch]), cl.df[gd], gacc[ch]);
            }
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) jac[gd * C + ch] = gacc[ch];
        }
    }
}

// ------------------------------------------------------------------ backward (float atomics)
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(kBlock) void grid_backward_atomic_kernel(
    const float *__restrict__ grad, const float *__restrict__ inputs, const int32_t *__restrict__ offsets,
    float *__restrict__ grad_table, uint32_t B, uint32_t nchunks, LevelRes lv, uint32_t gridtype,
    bool align_corners, uint32_t interp)
{
    const uint32_t item = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t level = item / nchunks;
    const uint32_t b = (item - level * nchunks) * kBlock + threadIdx.x;
    if (b >= B) return;

    const Geom<D> g = make_geom<D>(offsets, level, lv.res[level], gridtype);
    float *__restrict__ gt = grad_table + (size_t)(uint32_t)offsets[level] * C;

    float x[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
    Cell<D> cl;
    if (!locate<D>(x, g.res, align_corners, interp, cl)) return;

    Row<C> gr;
    gr.load(grad + ((size_t)level * B + b) * C);

#pragma unroll
    for (uint32_t corner = 0; corner < (1u << D); corner++) {
        float w = 1.0f;
        uint32_t c[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if (corner & (1u << d)) {
                w *= cl.f[d];
                c[d] = min(cl.c[d] + 1u, g.res - 1u);
            } else {
                w *= 1.0f - cl.f[d];
                c[d] = cl.c[d];
            }
        }
        float *dst = gt + (size_t)row_of<D>(g, c) * C;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) unsafeAtomicAdd(dst + ch, w * gr.v[ch]);
    }
}

// grad_inputs[b, d] = sum_{l, ch} grad[l, b, ch] * dy_dx[b, l, d, ch]   (all L levels)
template <uint32_t D, uint32_t C, bool LM = false>
__global__ __launch_bounds__(kBlock) void grid_input_backward_kernel(const float *__restrict__ grad,
                                                                     const float *__restrict__ dy_dx,
                                                                     float *__restrict__ grad_inputs, uint32_t B,
                                                                     uint32_t L)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float *jac = dy_dx + (size_t)b * L * D * C;
    float r = 0.0f;
    for (uint32_t l = 0; l < L; l++) {
        const float *gl = grad + ((size_t)l * B + b) * C;
        const float *jl = LM ? dy_dx + (((size_t)l * B + b) * D + d) * C : jac + (size_t)l * D * C + d * C;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) r = fmaf(gl[ch], jl[ch], r);
    }
    grad_inputs[t] = r;
}

// ------------------------------------------------------------------ total variation
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(kBlock) void grid_tv_kernel(const float *__restrict__ inputs,
                                                         const float *__restrict__ table, float *__restrict__ grad,
                                                         const int32_t *__restrict__ offsets, float w, uint32_t B,
                                                         uint32_t nchunks, LevelRes lv, uint32_t gridtype,
                                                         bool align_corners)
{
    const uint32_t item = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t level = item / nchunks;
    const uint32_t b = (item - level * nchunks) * kBlock + threadIdx.x;
    if (b >= B) return;
    const Geom<D> g = make_geom<D>(offsets, level, lv.res[level], gridtype);
    const float *__restrict__ tab = table + (size_t)(uint32_t)offsets[level] * C;
    float *__restrict__ ga = grad + (size_t)(uint32_t)offsets[level] * C;

    float x[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
    Cell<D> cl;
    if (!locate<D>(x, g.res, align_corners, 0u, cl)) return;

    uint32_t c[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) c[d] = cl.c[d];
    const uint32_t centre = row_of<D>(g, c);
    Row<C> mid;
    mid.load(tab + (size_t)centre * C);
    float sum[C], sq[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) sum[ch] = sq[ch] = 0.0f;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t cur = c[d];
        // + side is taken whenever cur < res, i.e. always: the neighbour index may equal res
        // (gridencoder.cu:595); it still lands inside the level through the hash / modulo.
        if (cur < g.res) {
            c[d] = cur + 1u;
            Row<C> nb;
            uint32_t r = row_of<D>(g, c);
            if (g.mode == 0) r %= g.T;  // a dense index with coordinate == res can reach T
            nb.load(tab + (size_t)r * C);
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) {
                const float dv = mid.v[ch] - nb.v[ch];
                sum[ch] += dv;
                sq[ch] = fmaf(dv, dv, sq[ch]);
            }
        }
        if (cur > 0u) {
            c[d] = cur - 1u;
            Row<C> nb;
            nb.load(tab + (size_t)row_of<D>(g, c) * C);
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) {
                const float dv = mid.v[ch] - nb.v[ch];
                sum[ch] += dv;
                sq[ch] = fmaf(dv, dv, sq[ch]);
            }
        }
        c[d] = cur;
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++)
        unsafeAtomicAdd(ga + (size_t)centre * C + ch, w * sum[ch] * (1.0f / sqrtf(sq[ch] + 1e-9f)));
}

// ------------------------------------------------------------------ weight decay
__global__ __launch_bounds__(kBlock) void grid_wd_kernel(const float *__restrict__ table, float *__restrict__ grad,
                                                         const int32_t *__restrict__ offsets, float weight,
                                                         uint32_t n_elems, uint32_t C, uint32_t L)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_elems; i += gridDim.x * kBlock) {
        const uint32_t n = i / C;
        uint32_t level = 0, lo = 0, hi = L;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)offsets[mid] <= n) {
                level = mid;
                lo = mid + 1;
            } else {
                hi = mid;
            }
        }
        const uint32_t T = (uint32_t)(offsets[level + 1] - offsets[level]);
        grad[i] += 2.0f * weight * table[i] / (float)T;
    }
}

// ------------------------------------------------------------------ host side
template <uint32_t D, uint32_t C>
static void launch_forward(const float *inputs, const float *table, const int32_t *offsets, float *outputs,
                           float *dy_dx, uint32_t B, uint32_t L, uint32_t max_level, const LevelRes &lv,
                           uint32_t gridtype, bool align, uint32_t interp, bool lm, hipStream_t st)
{
    const uint32_t nchunks = ceil_div(B, kBlock);
    const dim3 grid(snake_blocks(max_level, nchunks));
    if (dy_dx && lm)
        grid_forward_kernel<D, C, true, true><<<grid, kBlock, 0, st>>>(inputs, table, offsets, outputs, dy_dx, B, L,
                                                                       nchunks, max_level, lv, gridtype, align, interp);
    else if (dy_dx)
        grid_forward_kernel<D, C, true><<<grid, kBlock, 0, st>>>(inputs, table, offsets, outputs, dy_dx, B, L,
                                                                 nchunks, max_level, lv, gridtype, align, interp);
    else
        grid_forward_kernel<D, C, false><<<grid, kBlock, 0, st>>>(inputs, table, offsets, outputs, dy_dx, B, L,
                                                                  nchunks, max_level, lv, gridtype, align, interp);
}

template <uint32_t D, uint32_t C>
static void launch_backward(const float *grad, const float *inputs, const int32_t *offsets, float *grad_table,
                            uint32_t B, uint32_t L, uint32_t max_level, const LevelRes &lv, const float *dy_dx,
                            float *grad_inputs, uint32_t gridtype, bool align, uint32_t interp, hipStream_t st)
{
    const uint32_t nchunks = ceil_div(B, kBlock);
    grid_backward_atomic_kernel<D, C><<<dim3(nchunks * max_level), kBlock, 0, st>>>(
        grad, inputs, offsets, grad_table, B, nchunks, lv, gridtype, align, interp);
    if (dy_dx && grad_inputs)
        grid_input_backward_kernel<D, C><<<dim3(ceil_div(B * D, kBlock)), kBlock, 0, st>>>(grad, dy_dx, grad_inputs,
                                                                                          B, L);
}

template <uint32_t D, uint32_t C>
static void launch_tv(const float *inputs, const float *table, float *grad, const int32_t *offsets, float w,
                      uint32_t B, uint32_t L, const LevelRes &lv, uint32_t gridtype, bool align, hipStream_t st)
{
    const uint32_t nchunks = ceil_div(B, kBlock);
    grid_tv_kernel<D, C><<<dim3(nchunks * L), kBlock, 0, st>>>(inputs, table, grad, offsets, w, B, nchunks, lv,
                                                              gridtype, align);
}

// dispatch over the (D, C) pairs the reference instantiates (gridencoder.cu:385-410)
#define NGP_DISPATCH_C(D_, FN, ...)                         \
    switch (C) {                                            \
        case 1: FN<D_, 1>(__VA_ARGS__); break;              \
        case 2: FN<D_, 2>(__VA_ARGS__); break;              \
        case 4: FN<D_, 4>(__VA_ARGS__); break;              \
        case 8: FN<D_, 8>(__VA_ARGS__); break;              \
        case 16: FN<D_, 16>(__VA_ARGS__); break;            \
        case 32: FN<D_, 32>(__VA_ARGS__); break;            \
    }
#define NGP_DISPATCH_DC(FN, ...)                            \
    switch (D) {                                            \
        case 2: NGP_DISPATCH_C(2, FN, __VA_ARGS__) break;   \
        case 3: NGP_DISPATCH_C(3, FN, __VA_ARGS__) break;   \
        case 4: NGP_DISPATCH_C(4, FN, __VA_ARGS__) break;   \
        case 5: NGP_DISPATCH_C(5, FN, __VA_ARGS__) break;   \
    }

static int check_dc(const char *fn, uint32_t D, uint32_t C)
{
    // same messages as the reference's std::runtime_error (gridencoder.cu:392,409)
    if (!(C == 1 || C == 2 || C == 4 || C == 8 || C == 16 || C == 32))
        NGP_FAIL(NGP_EINVAL, "%s: GridEncoding: C must be 1, 2, 4, 8, 16 or 32.", fn);
    if (D < 2 || D > 5) NGP_FAIL(NGP_EINVAL, "%s: GridEncoding: D must be 2, 3, 4 or 5.", fn);
    return NGP_OK;
}

}  // namespace ngp

using namespace ngp;

extern "C" int ngp_grid_encode_forward(const float *inputs, const float *embeddings, const int32_t *offsets,
                                       float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                       uint32_t max_level, float S, uint32_t H, float *dy_dx, uint32_t gridtype,
                                       int align_corners, uint32_t interp, ngp_stream_t stream)
{
    if (B == 0 || max_level == 0) return NGP_OK;
    NGP_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward: null tensor");
    if (int e = check_dc("grid_encode_forward", D, C)) return e;
    LevelRes lv;
    NGP_REQUIRE(fill_levels(lv, S, H, L), "grid_encode_forward: L must be in [1, %u]", kMaxLevels);
    NGP_REQUIRE(max_level <= L, "grid_encode_forward: max_level > L");
    NGP_DISPATCH_DC(launch_forward, inputs, embeddings, offsets, outputs, dy_dx, B, L, max_level, lv, gridtype,
                    align_corners != 0, interp, false, as_stream(stream));
    NGP_CHECK_LAUNCH("grid_encode_forward");
    return NGP_OK;
}

extern "C" int ngp_x_grid_encode_forward_jac(const float *inputs, const float *embeddings, const int32_t *offsets,
                                             float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                             uint32_t max_level, float S, uint32_t H, float *dy_dx, uint32_t gridtype,
                                             int align_corners, uint32_t interp, int level_major, ngp_stream_t stream)
{
    if (B == 0 || max_level == 0) return NGP_OK;
    NGP_REQUIRE(inputs && embeddings && offsets && outputs && dy_dx, "grid_encode_forward_jac: null tensor");
    if (int e = check_dc("grid_encode_forward_jac", D, C)) return e;
    LevelRes lv;
    NGP_REQUIRE(fill_levels(lv, S, H, L), "grid_encode_forward_jac: L must be in [1, %u]", kMaxLevels);
    NGP_REQUIRE(max_level <= L, "grid_encode_forward_jac: max_level > L");
    NGP_DISPATCH_DC(launch_forward, inputs, embeddings, offsets, outputs, dy_dx, B, L, max_level, lv, gridtype,
                    align_corners != 0, interp, level_major != 0, as_stream(stream));
    NGP_CHECK_LAUNCH("grid_encode_forward_jac");
    return NGP_OK;
}

extern "C" int ngp_grid_encode_backward(const float *grad, const float *inputs, const float *embeddings,
                                        const int32_t *offsets, float *grad_embeddings, uint32_t B, uint32_t D,
                                        uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                        const float *dy_dx, float *grad_inputs, uint32_t gridtype,
                                        int align_corners, uint32_t interp, ngp_stream_t stream)
{
    if (B == 0) return NGP_OK;
    (void)embeddings;
    NGP_REQUIRE(grad && inputs && offsets && grad_embeddings, "grid_encode_backward: null tensor");
    if (int e = check_dc("grid_encode_backward", D, C)) return e;
    LevelRes lv;
    NGP_REQUIRE(fill_levels(lv, S, H, L), "grid_encode_backward: L must be in [1, %u]", kMaxLevels);
    NGP_REQUIRE(max_level <= L, "grid_encode_backward: max_level > L");
    NGP_DISPATCH_DC(launch_backward, grad, inputs, offsets, grad_embeddings, B, L, max_level, lv, dy_dx,
                    grad_inputs, gridtype, align_corners != 0, interp, as_stream(stream));
    NGP_CHECK_LAUNCH("grid_encode_backward");
    return NGP_OK;
}

template <uint32_t D, uint32_t C>
static void launch_input_backward(const float *grad, const float *dy_dx, float *grad_inputs, uint32_t B, uint32_t L,
                                  bool lm, hipStream_t st)
{
    const dim3 grid(ceil_div(B * D, kBlock));
    if (lm)
        grid_input_backward_kernel<D, C, true><<<grid, kBlock, 0, st>>>(grad, dy_dx, grad_inputs, B, L);
    else
        grid_input_backward_kernel<D, C, false><<<grid, kBlock, 0, st>>>(grad, dy_dx, grad_inputs, B, L);
}

extern "C" int ngp_x_grid_input_backward(const float *grad, const float *dy_dx, float *grad_inputs, uint32_t B,
                                         uint32_t D, uint32_t C, uint32_t L, int level_major, ngp_stream_t stream)
{
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(grad && dy_dx && grad_inputs, "grid_input_backward: null tensor");
    if (int e = check_dc("grid_input_backward", D, C)) return e;
    NGP_DISPATCH_DC(launch_input_backward, grad, dy_dx, grad_inputs, B, L, level_major != 0, as_stream(stream));
    NGP_CHECK_LAUNCH("grid_input_backward");
    return NGP_OK;
}

extern "C" int ngp_grad_total_variation(const float *inputs, const float *embeddings, float *grad,
                                        const int32_t *offsets, float weight, uint32_t B, uint32_t D, uint32_t C,
                                        uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                        ngp_stream_t stream)
{
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && embeddings && grad && offsets, "grad_total_variation: null tensor");
    if (int e = check_dc("grad_total_variation", D, C)) return e;
    LevelRes lv;
    NGP_REQUIRE(fill_levels(lv, S, H, L), "grad_total_variation: L must be in [1, %u]", kMaxLevels);
    const float w = weight / (float)(2u * D);
    NGP_DISPATCH_DC(launch_tv, inputs, embeddings, grad, offsets, w, B, L, lv, gridtype, align_corners != 0,
                    as_stream(stream));
    NGP_CHECK_LAUNCH("grad_total_variation");
    return NGP_OK;
}

extern "C" int ngp_grad_weight_decay(const float *embeddings, float *grad, const int32_t *offsets, float weight,
                                     uint32_t B, uint32_t C, uint32_t L, ngp_stream_t stream)
{
    const uint64_t n = (uint64_t)B * C;
    if (n == 0) return NGP_OK;
    NGP_REQUIRE(embeddings && grad && offsets, "grad_weight_decay: null tensor");
    NGP_REQUIRE(L >= 1, "grad_weight_decay: L must be >= 1");
    NGP_REQUIRE(n < (1ull << 32), "grad_weight_decay: table too large");
    const uint32_t blocks = min(ceil_div((uint32_t)n, kBlock), 256u * 16u);
    grid_wd_kernel<<<dim3(blocks), kBlock, 0, as_stream(stream)>>>(embeddings, grad, offsets, weight, (uint32_t)n, C, L);
    NGP_CHECK_LAUNCH("grad_weight_decay");
    return NGP_OK;
}
