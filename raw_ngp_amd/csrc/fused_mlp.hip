// Fused tiny-MLP field evaluation on the matrix cores (gfx950 MFMA, f16 operands, f32 accumulate).
//
// Replaces, for the default field configuration, what the reference does as six nn.Linear GEMMs plus
// slicing / cat / activation kernels per direction (nerf/network.py:27-35, :74-143; under `--fp16` those
// GEMMs run in fp16 autocast, renderer.py:546):
//     h     = W3 relu(W2 relu(W1 enc))                 enc = 32 hash-grid features
//     sigma = exp(h[0])                                  (trunc_exp, activation.py:9-19)
//     rgb   = min(exp(W6 relu(W5 relu(W4 [h[1:16], SH16(dir)])) - 5), 5)        (clamped_exp)
// One wave = 32 samples; the six layers are chained through MFMA accumulators (see mlp_common.hpp), so no
// activation ever leaves the register file in the forward pass; the encoder output is consumed in the
// level-major [L, stride, 2] slab layout the grid kernel writes (no permute), SH is evaluated in-kernel.
#include "binned_common.hpp"
#include "mlp_common.hpp"
#include "sh_eval.hpp"

namespace ngp {

__global__ __launch_bounds__(256) void mlp_prepare_kernel(MlpWeights W, _Float16 *__restrict__ image)
{
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kMlpImageHalfs) return;
    const uint32_t t = e & 7u, lane = (e >> 3) & 63u, f = e >> 9;
    image[e] = (_Float16)frag_elem(f, lane & 31u, lane >> 5, t, W);
}

// ------------------------------------------------------------------ per-tile pieces shared with the backward kernels
struct TileIn {
    half8 x0[2];   // encoder features as B fragments (k-steps 0, 1)
    half8 sh;      // SH16(dir) as the k-step-1 fragment of the view-MLP input
};

// lane (n, h): loads its half of the 32 encoder features of sample `row` and the SH basis of its direction
__device__ __forceinline__ TileIn load_tile(const float *__restrict__ enc, size_t stride, const float *__restrict__ dirs,
                                            uint32_t row, bool valid, uint32_t h)
{
    TileIn in;
#pragma unroll
    for (uint32_t s = 0; s < 2; s++)
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            // features 16s + 8q + 4h + {0..3} = levels 8s + 4q + 2h + {0, 1}, both channels
            const uint32_t level = 8 * s + 4 * q + 2 * h;
            float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
            if (valid) {
                a = reinterpret_cast<const float2 *>(enc)[(size_t)level * stride + row];
                b = reinterpret_cast<const float2 *>(enc)[(size_t)(level + 1) * stride + row];
            }
            in.x0[s][4 * q + 0] = (_Float16)a.x;
            in.x0[s][4 * q + 1] = (_Float16)a.y;
            in.x0[s][4 * q + 2] = (_Float16)b.x;
            in.x0[s][4 * q + 3] = (_Float16)b.y;
        }
    float dx = 0.f, dy = 0.f, dz = 1.f;
    if (valid && dirs) {
        dx = dirs[(size_t)row * 3];
        dy = dirs[(size_t)row * 3 + 1];
        dz = dirs[(size_t)row * 3 + 2];
    }
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    float sh[16], j0[1], j1[1], j2[1];
    sh_eval<4, false>(dx * inv, dy * inv, dz * inv, sh, j0, j1, j2);
    // element t of the k-step-1 fragment is SH index kperm(1, h, t) - 16 = 8 (t >> 2) + 4 h + (t & 3)
#pragma unroll
    for (uint32_t t = 0; t < 8; t++) {
        const float lo = sh[8 * (t >> 2) + (t & 3)], hi = sh[8 * (t >> 2) + 4 + (t & 3)];
        in.sh[t] = (_Float16)(h ? hi : lo);
    }
    return in;
}

// One 64-wide layer from a 64-feature input held as fragments xin[kb][s]:  acc[rb] = sum W[rb][kb][s] * xin[kb][s],
// the weight fragments read from LDS (local fragment ids)
#define NGP_LAYER64_LDS(acc, lds, lbase, xin)                                                              \
    do {                                                                                                   \
        _Pragma("unroll") for (int rb = 0; rb < 2; rb++) {                                                 \
            acc[rb] = zero16();                                                                            \
            _Pragma("unroll") for (int kb = 0; kb < 2; kb++)                                               \
                _Pragma("unroll") for (int s = 0; s < 2; s++)                                              \
                    acc[rb] = mfma(lds[((lbase) + rb * 4 + kb * 2 + s) * 64 + lane], xin[kb][s], acc[rb]); \
        }                                                                                                  \
    } while (0)

// ------------------------------------------------------------------ forward
// Where the weights live: the fragments of the two 64 x 64 layers and of the last one (20 of 32, 20 KiB) in LDS, the rest
// (W1, W3, W4: 48 registers) in registers.  With all 32 in registers the kernel needs 215 VGPRs -- two waves per SIMD when it
// has the GPU to itself, but only ONE next to the four waves x 56 registers of the side stream's march
// (march_const_step_kernel), and half of its 512 workgroups then run in a second round: 25 us beside the march against 12
// alone.  At 135 registers two waves fit beside the march.
// PASSENGER: workgroup 0 does the step's one-workgroup bookkeeping (binned_common.hpp: step_begin_block) instead of
// evaluating the field -- as a kernel of its own it sat on the step's critical path with a dependent-launch gap on top,
// although nothing consumes its results before the compositor
#ifndef NGP_FWD_WG_PER_CU
#define NGP_FWD_WG_PER_CU 3
#endif
constexpr uint32_t kFwdWgPerCu = NGP_FWD_WG_PER_CU;   // workgroups (4 waves) per CU the kernel is compiled and launched for
template <bool PASSENGER, bool SOFT = false>
__global__ __launch_bounds__(256, kFwdWgPerCu) void mlp_forward_kernel(const float *__restrict__ enc, uint32_t stride,
                                                            const float *__restrict__ dirs,
                                                            const int32_t *__restrict__ M_dev, uint32_t M_host,
                                                            const half8 *__restrict__ image,
                                                            float *__restrict__ sigma, float *__restrict__ rgb,
                                                            StepBegin begin, FieldAct act = FieldAct{},
                                                            const int32_t *__restrict__ scatter_cell = nullptr)
{
    if (PASSENGER && blockIdx.x == 0) {
        step_begin_block(begin);
        return;
    }
    const uint32_t first = PASSENGER ? 1u : 0u;
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5;
    const uint32_t wave = ((blockIdx.x - first) * 256u + threadIdx.x) >> 6, n_waves = ((gridDim.x - first) * 256u) >> 6;
    const uint32_t M = M_dev ? min((uint32_t)max(M_dev[0], 0), M_host) : M_host;
    const uint32_t n_tiles = (M + 31u) >> 5;

    __shared__ half8 lds_w[20 * 64];   // local 0..7 = F_W2, 8..15 = F_W5, 16..19 = F_W6
    constexpr uint32_t L_W2 = 0, L_W5 = 8, L_W6 = 16;
    for (uint32_t i = threadIdx.x; i < 20u * 64u; i += 256u) {
        const uint32_t f = i >> 6;
        lds_w[i] = image[(size_t)(f < 8u ? F_W2 + f : (f < 16u ? F_W5 + (f - 8u) : F_W6 + (f - 16u))) * 64 + (i & 63u)];
    }
    half8 wf[20];   // F_W1 (0..3), F_W3 (12..15), F_W4 (16..19) stay in registers; the slots in between are never touched
#pragma unroll
    for (int i = 0; i < 20; i++)
        if (i < 4 || i >= 12) wf[i] = image[(size_t)i * 64 + lane];
    __syncthreads();

    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        asm volatile("" ::: "memory");   // keep the LDS fragments in LDS: no hoisting of 80 registers' worth out of the loop
        const uint32_t row = tile * 32u + n;
        const bool valid = row < M;
        const TileIn in = load_tile(enc, stride, dirs, row, valid, h);

        f32x16 a[2];
        half8 x[2][2];
        // layer 1: 32 -> 64
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            a[rb] = zero16();
#pragma unroll
            for (int s = 0; s < 2; s++) a[rb] = mfma(wf[F_W1 + rb * 2 + s], in.x0[s], a[rb]);
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            x[kb][0] = pack_hidden<0, SOFT>(a[kb], act.beta);
            x[kb][1] = pack_hidden<1, SOFT>(a[kb], act.beta);
        }
        // layer 2: 64 -> 64
        NGP_LAYER64_LDS(a, lds_w, L_W2, x);
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            x[kb][0] = pack_hidden<0, SOFT>(a[kb], act.beta);
            x[kb][1] = pack_hidden<1, SOFT>(a[kb], act.beta);
        }
        // layer 3: 64 -> 16 (rows 0..15 of one tile): row 0 = raw density, rows 1..15 = geometry features
        f32x16 o = zero16();
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int s = 0; s < 2; s++) o = mfma(wf[F_W3 + kb * 2 + s], x[kb][s], o);
        const float sigma_raw = o[0];
        if (rgb == nullptr) {   // density-only query (density-grid refresh): skip the colour MLP
            if (valid && h == 0) {
                const float value = act_sigma(sigma_raw, act);
                if (scatter_cell) {   // density-grid refresh: `sigma` is the cascade's scratch grid (starts at -1), max per cell.
                    // sigma >= 0: as signed integers, float bit patterns of non-negative values order like the floats
                    const int32_t cell = scatter_cell[row];
                    if (cell >= 0) atomicMax(reinterpret_cast<int *>(sigma) + cell, __float_as_int(value));
                } else {
                    sigma[row] = value;
                }
            }
            continue;
        }
        // layer 4: [sigma row (zero weight), 15 features, SH16] -> 64
        const half8 x3a = pack<0, false>(o);
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            a[rb] = zero16();
            a[rb] = mfma(wf[F_W4 + rb * 2 + 0], x3a, a[rb]);
            a[rb] = mfma(wf[F_W4 + rb * 2 + 1], in.sh, a[rb]);
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            x[kb][0] = pack_hidden<0, SOFT>(a[kb], act.beta);
            x[kb][1] = pack_hidden<1, SOFT>(a[kb], act.beta);
        }
        // layer 5: 64 -> 64
        NGP_LAYER64_LDS(a, lds_w, L_W5, x);
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            x[kb][0] = pack_hidden<0, SOFT>(a[kb], act.beta);
            x[kb][1] = pack_hidden<1, SOFT>(a[kb], act.beta);
        }
        // layer 6: 64 -> 3
        f32x16 c = zero16();
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int s = 0; s < 2; s++) c = mfma(lds_w[(L_W6 + kb * 2 + s) * 64 + lane], x[kb][s], c);

        if (valid && h == 0) {   // rows 0..3 of a tile live in registers 0..3 of the h = 0 lanes
            sigma[row] = act_sigma(sigma_raw, act);
            rgb[(size_t)row * 3 + 0] = act_color(c[0], act);
            rgb[(size_t)row * 3 + 1] = act_color(c[1], act);
            rgb[(size_t)row * 3 + 2] = act_color(c[2], act);
        }
    }
}

}  // namespace ngp

using namespace ngp;

extern "C" size_t ngp_x_mlp_image_bytes(void) { return (size_t)kMlpImageHalfs * 2; }

extern "C" int ngp_x_mlp_prepare(const float *w1, const float *w2, const float *w3, const float *w4, const float *w5,
                                 const float *w6, void *image, ngp_stream_t stream)
{
    NGP_REQUIRE(w1 && w2 && w3 && w4 && w5 && w6 && image, "mlp_prepare: null tensor");
    NGP_REQUIRE(((uintptr_t)image & 15u) == 0, "mlp_prepare: image must be 16-byte aligned");
    MlpWeights W{w1, w2, w3, w4, w5, w6};
    mlp_prepare_kernel<<<dim3(ceil_div(kMlpImageHalfs, 256u)), dim3(256), 0, as_stream(stream)>>>(
        W, reinterpret_cast<_Float16 *>(image));
    NGP_CHECK_LAUNCH("mlp_prepare");
    return NGP_OK;
}

extern "C" int ngp_x_mlp_forward(const float *enc, uint32_t stride, const float *dirs, const int32_t *M_dev, uint32_t M,
                                 const void *image, float *sigma, float *rgb, ngp_stream_t stream)
{
    return ngp_x_mlp_forward_act(enc, stride, dirs, M_dev, M, image, sigma, rgb, 0, 0, 0, 1.0f, stream);
}

// ... with the field's non-default output activations (network.py:111-135): color_act 0 clamped_exp / 1 exp / 2 sigmoid,
// density_act 0 trunc_exp / 1 softplus(beta, threshold 20)
extern "C" int ngp_x_mlp_forward_act(const float *enc, uint32_t stride, const float *dirs, const int32_t *M_dev, uint32_t M,
                                     const void *image, float *sigma, float *rgb, uint32_t color_act, uint32_t density_act,
                                     uint32_t internal_act, float beta, ngp_stream_t stream)
{
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(color_act <= 2u && density_act <= 1u && internal_act <= 1u && beta > 0.0f,
                "mlp_forward: unknown activation or beta <= 0");
    NGP_REQUIRE(enc && image && sigma, "mlp_forward: null tensor");
    NGP_REQUIRE(rgb == nullptr || dirs != nullptr, "mlp_forward: dirs missing");
    NGP_REQUIRE(stride >= M, "mlp_forward: encoder slab stride smaller than M");
    const uint32_t tiles = ceil_div(M, 32u);
    const uint32_t blocks = min(ceil_div(tiles, 4u), 256u * kFwdWgPerCu);
    FieldAct act;
    act.color = color_act, act.density = density_act, act.beta = beta, act.internal = internal_act;
    if (internal_act)
        mlp_forward_kernel<false, true><<<dim3(blocks), dim3(256), 0, as_stream(stream)>>>(
            enc, stride, dirs, M_dev, M, reinterpret_cast<const half8 *>(image), sigma, rgb, StepBegin{}, act);
    else
        mlp_forward_kernel<false><<<dim3(blocks), dim3(256), 0, as_stream(stream)>>>(
            enc, stride, dirs, M_dev, M, reinterpret_cast<const half8 *>(image), sigma, rgb, StepBegin{}, act);
    NGP_CHECK_LAUNCH("mlp_forward");
    return NGP_OK;
}

// Density only, scattered: the refresh's evaluate + ngp_x_density_grid_scatter as one launch.  Row i's density goes to
// tmp_cas[cells[i]] by atomic max (cells[i] < 0: dropped), instead of to a sigma array that a second kernel would re-read.
extern "C" int ngp_x_mlp_density_scatter(const float *enc, uint32_t stride, uint32_t M, const void *image,
                                         const int32_t *cells, float *tmp_cas, uint32_t density_act, uint32_t internal_act,
                                         float beta, ngp_stream_t stream)
{
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(density_act <= 1u && internal_act <= 1u && beta > 0.0f, "mlp_density_scatter: unknown activation or beta <= 0");
    NGP_REQUIRE(enc && image && cells && tmp_cas, "mlp_density_scatter: null tensor");
    NGP_REQUIRE(stride >= M, "mlp_density_scatter: encoder slab stride smaller than M");
    const uint32_t tiles = ceil_div(M, 32u);
    const uint32_t blocks = min(ceil_div(tiles, 4u), 256u * kFwdWgPerCu);
    FieldAct act;
    act.color = 0, act.density = density_act, act.beta = beta, act.internal = internal_act;
    if (internal_act)
        mlp_forward_kernel<false, true><<<dim3(blocks), dim3(256), 0, as_stream(stream)>>>(
            enc, stride, nullptr, nullptr, M, reinterpret_cast<const half8 *>(image), tmp_cas, nullptr, StepBegin{}, act, cells);
    else
        mlp_forward_kernel<false><<<dim3(blocks), dim3(256), 0, as_stream(stream)>>>(
            enc, stride, nullptr, nullptr, M, reinterpret_cast<const half8 *>(image), tmp_cas, nullptr, StepBegin{}, act, cells);
    NGP_CHECK_LAUNCH("mlp_density_scatter");
    return NGP_OK;
}

// ngp_x_mlp_forward with ngp_x_step_begin (same arguments) as one more workgroup of the same launch
extern "C" int ngp_x_mlp_forward_step_begin(const float *enc, uint32_t stride, const float *dirs, const int32_t *M_dev,
                                            uint32_t M, const void *image, float *sigma, float *rgb, uint32_t *step_counter,
                                            float *hyper, double lr0, double decay_steps, double beta1, double beta2,
                                            float *loss_out, int64_t *samples_seen, const int32_t *sample_counter,
                                            void *binned_workspace, uint32_t L, uint32_t n_rows_total, int single_segment,
                                            float *scaler, double growth, double backoff, uint32_t growth_interval,
                                            ngp_stream_t stream)
{
    NGP_REQUIRE(M != 0, "mlp_forward_step_begin: M must be positive");
    NGP_REQUIRE(enc && image && sigma, "mlp_forward_step_begin: null tensor");
    NGP_REQUIRE(rgb == nullptr || dirs != nullptr, "mlp_forward_step_begin: dirs missing");
    NGP_REQUIRE(stride >= M, "mlp_forward_step_begin: encoder slab stride smaller than M");
    StepBegin a;
    const int rc = step_begin_args(a, "mlp_forward_step_begin", step_counter, hyper, lr0, decay_steps, beta1, beta2, loss_out,
                                   samples_seen, sample_counter, binned_workspace, L, n_rows_total, single_segment, scaler,
                                   growth, backoff, growth_interval);
    if (rc != NGP_OK) return rc;
    const uint32_t tiles = ceil_div(M, 32u);
    const uint32_t blocks = min(ceil_div(tiles, 4u), 256u * kFwdWgPerCu);
    mlp_forward_kernel<true><<<dim3(blocks + 1), dim3(256), 0, as_stream(stream)>>>(
        enc, stride, dirs, M_dev, M, reinterpret_cast<const half8 *>(image), sigma, rgb, a);
    NGP_CHECK_LAUNCH("mlp_forward_step_begin");
    return NGP_OK;
}
