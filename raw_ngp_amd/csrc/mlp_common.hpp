// Shared definitions of the fused tiny-MLP kernels (fused_mlp_*.hip): MFMA operand fragments.
//
// The field's two bias-free MLPs (nerf/network.py:49,56 of the reference: 32-64-64-16 and 31-64-64-3,
// ReLU) are evaluated with v_mfma_f32_32x32x16_f16, activations as [feature rows x sample columns]
// accumulator tiles: one wave = 32 samples (columns on lanes), features in registers.
//
// MFMA 32x32x16 register maps (lane l: r = l & 31, h = l >> 5):
//   A operand   element t (0..7) = A[row r][k = 8h + t]
//   B operand   element t        = B[k = 8h + t][col r]
//   C / D       register v (0..15) = D[row (v&3) + 8(v>>2) + 4h][col r]
// Feeding registers 8s..8s+7 of a D tile X as the next MFMA's B operand contracts over X's rows with
//   logical k = kperm(s, h, t) = 16 s + 8 (t >> 2) + 4 h + (t & 3)           (s = k-step inside the 32-row tile)
// and feeding them as the A operand gives X^T (rows = X's columns).  Every weight fragment below is
// therefore stored as frag[lane (r, h)][t] = M[r][kperm(s, h, t)] for a 32 x 32 block M of the
// (possibly transposed, zero-padded) weight matrix; the same bits serve as A-of-M or B-of-M^T.
#pragma once
#include "ngp_common.hpp"

namespace ngp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr uint32_t kperm(uint32_t s, uint32_t h, uint32_t t)
{
    return 16u * s + 8u * (t >> 2) + 4u * h + (t & 3u);
}

// Fragment ids inside the f16 weight image (each fragment = 64 lanes x 8 halfs = 1 KiB).
enum MlpFrag : uint32_t {
    F_W1 = 0,    // [rb 2][s 2]            W1[32rb + r][k]
    F_W2 = 4,    // [rb 2][kb 2][s 2]      W2[32rb + r][32kb + k]
    F_W3 = 12,   // [kb 2][s 2]            r < 16 ? W3[r][32kb + k] : 0
    F_W4 = 16,   // [rb 2][s 2]            k >= 1 ? W4[32rb + r][k - 1] : 0     (k = 0 is the sigma row)
    F_W5 = 20,   // [rb 2][kb 2][s 2]
    F_W6 = 28,   // [kb 2][s 2]            r < 3 ? W6[r][32kb + k] : 0
    T_W6 = 32,   // [rb 2]        (s = 0)  k < 3 ? W6[k][32rb + r] : 0
    T_W5 = 34,   // [rb 2][kb 2][s 2]      W5[32kb + k][32rb + r]
    T_W4 = 42,   // [kb 2][s 2]            r >= 1 ? W4[32kb + k][r - 1] : 0
    T_W3 = 46,   // [rb 2]        (s = 0)  k < 16 ? W3[k][32rb + r] : 0
    T_W2 = 48,   // [rb 2][kb 2][s 2]      W2[32kb + k][32rb + r]
    T_W1 = 56,   // [kb 2][s 2]            W1[32kb + k][r]
    kMlpFrags = 60,
};
constexpr uint32_t kMlpImageHalfs = kMlpFrags * 64 * 8;   // 30720 halfs = 60 KiB

// weight matrix shapes (torch layout [out][in], fp32)
constexpr uint32_t kW1 = 64 * 32, kW2 = 64 * 64, kW3 = 16 * 64, kW4 = 64 * 31, kW5 = 64 * 64, kW6 = 3 * 64;
constexpr uint32_t kMlpParams = kW1 + kW2 + kW3 + kW4 + kW5 + kW6;   // 13440

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16()
{
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    return z;
}

// registers 8S..8S+7 of an accumulator tile as an f16 operand fragment
template <int S, bool RELU>
__device__ __forceinline__ half8 pack(const f32x16 &a)
{
    half8 o;
#pragma unroll
    for (int t = 0; t < 8; t++) o[t] = (_Float16)a[8 * S + t];
    // ReLU after the conversion (same result: the conversion is monotone and keeps 0) as 4 packed v_pk_max_f16
    // instead of 8 v_max_f32
    if (RELU) o = __builtin_elementwise_max(o, (half8)(_Float16)0.0f);
    return o;
}

// conversion of a delta tile to f16 with saturation at +-lim: a plain cast turns |x| > 65504 into inf, and one inf in a
// delta becomes NaN weights for good.  Static loss scale: lim = 65504, the sample's gradient is clipped.  Dynamic loss
// scale (LossScaler below): lim = inf, the overflow is let through ON PURPOSE -- it poisons the weight gradient of its
// own layer, the reduction of the weight gradients sees it, the step is skipped and the scale halved, which is what the
// reference's GradScaler does (train_utils.py:404,897-904).  Same instruction count either way.
template <int S>
__device__ __forceinline__ half8 pack_sat(const f32x16 &a, _Float16 lim)
{
    half8 o = pack<S, false>(a);
    o = __builtin_elementwise_min(__builtin_elementwise_max(o, (half8)(-lim)), (half8)lim);
    return o;
}
// ---- dynamic loss scale (torch.cuda.amp.GradScaler on the device) ----------------------------------------------------
// eight 32-bit words in HBM, owned by the caller: [0] f32 scale, [1] f32 1 / scale, [2] u32 "a non-finite gradient was seen
// in the current step", [3] u32 clean steps since the scale last changed, [4] u32 optimiser steps taken (Adam's t),
// [5] u32 steps skipped, [6] u32 a step has run since the words were last settled, [7] reserved.
// Readers: the MLP backward kernels ([0], [1]); writers of [2]: the weight-gradient reduction and the table reduce;
// the optimiser kernels skip on [2]; step_begin settles the previous step (binned_common.hpp: step_begin_block).
enum LossScalerWord : uint32_t { LS_SCALE = 0, LS_INV = 1, LS_FOUND = 2, LS_TRACKER = 3, LS_ADAM_T = 4, LS_SKIPPED = 5, LS_PENDING = 6 };
__device__ __forceinline__ _Float16 delta_limit(const float *scaler)
{
    return scaler ? (_Float16)__uint_as_float(0x7f800000u) : (_Float16)65504.0f;
}
struct MlpWeights {
    const float *w1, *w2, *w3, *w4, *w5, *w6;
};

// ---- output activations of the field (nerf/network.py:111-135 of the reference) ------------------------------------------
// density: 0 = trunc_exp (exp forward, exp(clamp(x, -80?, 80)) backward: activation.py; `clamped_exp`, the default),
//          1 = softplus(beta, threshold 20);   colour: 0 = clamp(exp(x - 5), max 5) (default), 1 = exp(x - 5), 2 = sigmoid
//          hidden layers (`internal`): 0 = ReLU (default), 1 = softplus(beta, threshold 20)   (network.py:31-34)
struct FieldAct {
    uint32_t color = 0, density = 0;
    float beta = 1.0f;
    uint32_t internal = 0;
};
// hidden activation of registers 8S..8S+7 of an accumulator tile, as an f16 operand fragment.  SOFT: F.softplus(x, beta,
// threshold 20) evaluated in f32 before the conversion (the ReLU form clamps after it: four packed ops)
template <int S, bool SOFT>
__device__ __forceinline__ half8 pack_hidden(const f32x16 &a, float beta)
{
    if constexpr (!SOFT) {
        return pack<S, true>(a);
    } else {
        half8 o;
        const float inv = 1.0f / beta;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const float x = a[8 * S + t], z = beta * x;
            o[t] = (_Float16)(z > 20.0f ? x : log1pf(__expf(z)) * inv);
        }
        return o;
    }
}
// delta through a hidden activation: registers 8S..8S+7 of `a` times the activation's derivative, taken from the matching
// post-activation fragment `act` -- ReLU: zero where act is 0 (a bit mask); softplus: sigmoid(beta x) = 1 - exp(-beta
// softplus(x)) (1 in the linear region, where exp(-beta x) < 2e-9) -- then the saturating conversion of pack_sat
template <int S, bool SOFT>
__device__ __forceinline__ half8 pack_delta(const f32x16 &a, const half8 &act, _Float16 lim, float beta)
{
    if constexpr (!SOFT) {
        // mask = all ones where act > 0, built with packed 16-bit integer ops on the f16 bit patterns (a positive f16 is
        // a positive int16; -0 is negative): clamp to {0, 1}, multiply by 0xffff
        typedef short short8 __attribute__((ext_vector_type(8)));
        const half8 o = pack_sat<S>(a, lim);
        short8 m = __builtin_bit_cast(short8, act);
        m = __builtin_elementwise_min(__builtin_elementwise_max(m, (short8)0), (short8)1) * (short8)-1;
        return __builtin_bit_cast(half8, (short8)(__builtin_bit_cast(short8, o) & m));
    } else {
        f32x16 d = a;
#pragma unroll
        for (int t = 0; t < 8; t++) d[8 * S + t] = a[8 * S + t] * (1.0f - __expf(-beta * (float)act[t]));
        return pack_sat<S>(d, lim);
    }
}
__device__ __forceinline__ float act_sigma(float h, const FieldAct &a)
{
    if (a.density == 0u) return __expf(h);
    const float z = a.beta * h;                                  // F.softplus(h, beta, threshold = 20)
    return z > 20.0f ? h : log1pf(__expf(z)) / a.beta;
}
__device__ __forceinline__ float act_dsigma(float h, const FieldAct &a)   // d sigma / d h
{
    if (a.density == 0u) return __expf(fminf(fmaxf(h, -80.0f), 80.0f));   // trunc_exp's backward
    const float z = a.beta * h;
    return z > 20.0f ? 1.0f : 1.0f / (1.0f + __expf(-z));
}
__device__ __forceinline__ float act_color(float c, const FieldAct &a)
{
    if (a.color == 2u) return 1.0f / (1.0f + __expf(-c));
    const float e = __expf(c - 5.0f);
    return a.color == 1u ? e : fminf(e, 5.0f);
}
__device__ __forceinline__ float act_dcolor(float c, const FieldAct &a)   // d colour / d c
{
    if (a.color == 2u) {
        const float s = 1.0f / (1.0f + __expf(-c));
        return s * (1.0f - s);
    }
    const float e = __expf(c - 5.0f);
    return (a.color == 1u || e <= 5.0f) ? e : 0.0f;              // (the clamp's gradient is zero where it is active)
}

// ------------------------------------------------------------------ weight image
__device__ __forceinline__ float frag_elem(uint32_t f, uint32_t r, uint32_t h, uint32_t t, const MlpWeights &W)
{
    if (f < F_W2) {
        const uint32_t i = f - F_W1, rb = i >> 1, s = i & 1;
        return W.w1[(32 * rb + r) * 32 + kperm(s, h, t)];
    }
    if (f < F_W3) {
        const uint32_t i = f - F_W2, rb = i >> 2, kb = (i >> 1) & 1, s = i & 1;
        return W.w2[(32 * rb + r) * 64 + 32 * kb + kperm(s, h, t)];
    }
    if (f < F_W4) {
        const uint32_t i = f - F_W3, kb = i >> 1, s = i & 1;
        return r < 16 ? W.w3[r * 64 + 32 * kb + kperm(s, h, t)] : 0.0f;
    }
    if (f < F_W5) {
        const uint32_t i = f - F_W4, rb = i >> 1, s = i & 1, k = kperm(s, h, t);
        return k >= 1 ? W.w4[(32 * rb + r) * 31 + k - 1] : 0.0f;
    }
    if (f < F_W6) {
        const uint32_t i = f - F_W5, rb = i >> 2, kb = (i >> 1) & 1, s = i & 1;
        return W.w5[(32 * rb + r) * 64 + 32 * kb + kperm(s, h, t)];
    }
    if (f < T_W6) {
        const uint32_t i = f - F_W6, kb = i >> 1, s = i & 1;
        return r < 3 ? W.w6[r * 64 + 32 * kb + kperm(s, h, t)] : 0.0f;
    }
    if (f < T_W5) {
        const uint32_t rb = f - T_W6, k = kperm(0, h, t);
        return k < 3 ? W.w6[k * 64 + 32 * rb + r] : 0.0f;
    }
    if (f < T_W4) {
        const uint32_t i = f - T_W5, rb = i >> 2, kb = (i >> 1) & 1, s = i & 1;
        return W.w5[(32 * kb + kperm(s, h, t)) * 64 + 32 * rb + r];
    }
    if (f < T_W3) {
        const uint32_t i = f - T_W4, kb = i >> 1, s = i & 1;
        return r >= 1 ? W.w4[(32 * kb + kperm(s, h, t)) * 31 + r - 1] : 0.0f;
    }
    if (f < T_W2) {
        const uint32_t rb = f - T_W3, k = kperm(0, h, t);
        return k < 16 ? W.w3[k * 64 + 32 * rb + r] : 0.0f;
    }
    if (f < T_W1) {
        const uint32_t i = f - T_W2, rb = i >> 2, kb = (i >> 1) & 1, s = i & 1;
        return W.w2[(32 * kb + kperm(s, h, t)) * 64 + 32 * rb + r];
    }
    const uint32_t i = f - T_W1, kb = i >> 1, s = i & 1;
    return W.w1[(32 * kb + kperm(s, h, t)) * 32 + r];
}

// inverse of the map above: where element M[r][k] (k < 32) of the 32 x 32 block whose first fragment is `f` lives in
// the image (index in halfs); the block's second k-step, when it has one, is fragment f + 1
__host__ __device__ constexpr uint32_t frag_pos(uint32_t f, uint32_t r, uint32_t k)
{
    const uint32_t s = k >> 4, h = (k >> 2) & 1u, t = ((k >> 3) & 1u) * 4u + (k & 3u);
    return ((f + s) * 64u + r + 32u * h) * 8u + t;
}

// BARF level window (network.py:99-109 of the reference: f * w before the density MLP): the weights of one lane's 8
// encoder levels in load order (s, q, e): level = 8s + 4q + 2h + e
struct LaneWindow {
    float w[8];
};
__device__ __forceinline__ LaneWindow load_window(const float *__restrict__ level_w, uint32_t h)
{
    LaneWindow lw;
#pragma unroll
    for (uint32_t s = 0; s < 2; s++)
#pragma unroll
        for (uint32_t q = 0; q < 2; q++)
#pragma unroll
            for (uint32_t e = 0; e < 2; e++) lw.w[4 * s + 2 * q + e] = level_w ? level_w[8 * s + 4 * q + 2 * h + e] : 1.0f;
    return lw;
}

// Weight-gradient accumulators of a workgroup's four waves -> one partial slab in global memory, summed in wave order
// ((w0 + w1) + w2) + w3: a fixed order, so the result is bitwise reproducible.  Every wave parks its tiles in LDS at once
// (16-byte stores, its own region), one barrier, then all 256 lanes add the four copies and store the slab; done in two
// rounds of NT / 2 tiles so that 4 x NT/2 x 4 KiB of LDS suffice (64 KiB for 8 tiles, 80 KiB for 9).
// (The first version took turns -- wave k adds its tiles to the image while the other three wait at a barrier, four
// times -- and cost 24 000 cycles per wave, 43 % of the view kernel's run time: in-kernel stamps, tools/mlp_stamps.py.)
// slab element (tile b, register v, lane l) at (b * 16 + v) * 64 + l, as the reduction kernels expect.
template <int NT>
__device__ __forceinline__ void flush_tiles_parallel(float *lds_acc, const f32x16 (&g)[NT], uint32_t lane,
                                                     float *__restrict__ slab)
{
    constexpr int HALF = (NT + 1) / 2;
    const uint32_t wid = threadIdx.x >> 6;
    float4 *mine = reinterpret_cast<float4 *>(lds_acc) + (size_t)wid * HALF * 256;
    const float4 *all = reinterpret_cast<const float4 *>(lds_acc);
#pragma unroll
    for (int round = 0; round < 2; round++) {
        const int b0 = round * HALF, nb = round == 0 ? HALF : NT - HALF;
        __syncthreads();   // round 0: everyone is done reading weight fragments from this LDS; round 1: round 0 was read
#pragma unroll
        for (int b = 0; b < nb; b++)
#pragma unroll
            for (int vq = 0; vq < 4; vq++)
                mine[(b * 4 + vq) * 64 + lane] = make_float4(g[b0 + b][4 * vq], g[b0 + b][4 * vq + 1], g[b0 + b][4 * vq + 2],
                                                             g[b0 + b][4 * vq + 3]);
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < (uint32_t)nb * 256u; q += 256) {
            float4 s = all[q];
#pragma unroll
            for (uint32_t k = 1; k < 4; k++) {
                const float4 t = all[(size_t)k * HALF * 256 + q];
                s.x += t.x;
                s.y += t.y;
                s.z += t.z;
                s.w += t.w;
            }
            const uint32_t b = b0 + (q >> 8), vq = (q >> 6) & 3u, l = q & 63u;
            float *dst = slab + ((size_t)b * 16 + 4 * vq) * 64 + l;
            dst[0] = s.x;
            dst[64] = s.y;
            dst[128] = s.z;
            dst[192] = s.w;
        }
    }
}
constexpr size_t flush_lds_bytes(int NT) { return (size_t)4 * ((NT + 1) / 2) * 4096; }

// the density MLP's backward kernel (fused_mlp_backward.hip) for callers in other translation units: level_w may be
// NULL (no window); t3_base = first of the 14 transposed fragments (T_W3, T_W2, T_W1) in `image`
int launch_mlp_backward_grid(const float *enc, uint32_t stride, const float *level_w, const int32_t *M_dev, uint32_t M,
                             const half8 *image, uint32_t t3_base, float inv_loss_scale, const half8 *d3buf, float *denc,
                             float *partial, uint32_t blocks, hipStream_t st, const int32_t *sample_index = nullptr,
                             const float *scaler = nullptr, FieldAct act = FieldAct{});
// ... with a unit delta on the first output and no weight gradients: denc <- d h0 / d enc (ngp_x_mlp_density_gradient)
int launch_mlp_density_gradient(const char *who, const float *enc, uint32_t stride, const float *level_w, const int32_t *M_dev,
                                uint32_t M, const void *image, uint32_t t3_base, float *denc, hipStream_t st);


// ---- weight-gradient reduction over the backward kernels' partial slabs ----------------------------------------------
// element e of a workgroup slab: tile b = e / 1024, register v = (e / 64) % 16, lane = e % 64
// -> dW[32*rb + o][32*cb + j] with o = (v&3) + 8(v>>2) + 4(lane>>5), j = lane & 31
// optional: torch.optim.Adam (as engine_kernels.hip: adam_span) on the flat MLP weight buffer, element by element as the
// gradients come out of the reduction; dw1..dw6 must then be views of `grad`
constexpr uint32_t kAccFloats = 8 * 16 * 64;   // 8 accumulator tiles per kernel = 8192 floats per workgroup
constexpr uint32_t kDwGroups = 2 * kAccFloats / 64u;   // groups of 64 outputs (view kernel's slab, then the grid kernel's)
struct MlpAdam {
    float *param;
    const float *grad;
    float *exp_avg, *exp_avg_sq;
    const float *hyper;   // {lr, 1 - beta1^t, 1/sqrt(1 - beta2^t)}
    float b1, b2, eps;
    _Float16 *image;      // optional: the f16 operand image (ngp_x_mlp_prepare) is patched with the new weight
};
struct MlpDwReduce {
    const float *part_view, *part_grid;
    uint32_t n_wg;
    float inv_loss_scale;
    float *dw1, *dw2, *dw3, *dw4, *dw5, *dw6;
    MlpAdam adam;
    float *scaler = nullptr;   // dynamic loss scale (LossScalerWord): 1 / scale is read from it, non-finite sums raise LS_FOUND
};
int mlp_dw_reduce_args(MlpDwReduce &r, const char *who, uint32_t M, float loss_scale, float *dw1, float *dw2, float *dw3,
                       float *dw4, float *dw5, float *dw6, const void *workspace, size_t workspace_bytes, float *adam_param,
                       const float *adam_grad, float *adam_exp_avg, float *adam_exp_avg_sq, uint32_t adam_n,
                       const float *adam_hyper, float beta1, float beta2, float eps, void *adam_image, float *scaler = nullptr);

// element e (0 .. 2 kAccFloats) of the two partial slabs -> its weight (NULL: a padding element of the tiles) and the
// weight's two places in the f16 operand image (as-is and transposed block)
struct DwPlace {
    float *dst;
    uint32_t pos_f, pos_t;
};
__device__ __forceinline__ DwPlace mlp_dw_place(const MlpDwReduce &a, uint32_t e)
{
    const bool view = e >= kAccFloats;
    const uint32_t i = view ? e - kAccFloats : e;
    const uint32_t b = i >> 10, v = (i >> 6) & 15u, lane = i & 63u;
    const uint32_t o = (v & 3u) + 8u * (v >> 2) + 4u * (lane >> 5), j = lane & 31u;
    DwPlace p{nullptr, 0, 0};
    if (b < 2) {   // first-layer tiles: rb = b
        if (view) {
            if (j >= 1) {
                p.dst = &a.dw4[(32 * b + o) * 31 + j - 1];
                p.pos_f = frag_pos(F_W4 + b * 2, o, j);
                p.pos_t = frag_pos(T_W4 + b * 2, j, o);
            }
        } else {
            p.dst = &a.dw1[(32 * b + o) * 32 + j];
            p.pos_f = frag_pos(F_W1 + b * 2, o, j);
            p.pos_t = frag_pos(T_W1 + b * 2, j, o);
        }
    } else if (b < 6) {   // 64 x 64 tiles: rb = (b-2) >> 1, cb = (b-2) & 1
        const uint32_t rb = (b - 2) >> 1, cb = (b - 2) & 1;
        p.dst = &(view ? a.dw5 : a.dw2)[(32 * rb + o) * 64 + 32 * cb + j];
        p.pos_f = frag_pos((view ? F_W5 : F_W2) + rb * 4 + cb * 2, o, j);
        p.pos_t = frag_pos((view ? T_W5 : T_W2) + cb * 4 + rb * 2, j, o);
    } else {   // last-layer tiles: cb = b - 6
        const uint32_t cb = b - 6;
        if (view) {
            if (o < 3) {
                p.dst = &a.dw6[o * 64 + 32 * cb + j];
                p.pos_f = frag_pos(F_W6 + cb * 2, o, j);
                p.pos_t = frag_pos(T_W6 + cb, j, o);
            }
        } else {
            if (o < 16) {
                p.dst = &a.dw3[o * 64 + 32 * cb + j];
                p.pos_f = frag_pos(F_W3 + cb * 2, o, j);
                p.pos_t = frag_pos(T_W3 + cb, j, o);
            }
        }
    }
    return p;
}
// torch.optim.Adam on one MLP weight whose gradient is s (+ the weight's two entries in the f16 operand image)
__device__ __forceinline__ void mlp_adam_element(const MlpAdam &adam, const DwPlace &pl, float s)
{
    const size_t k = (size_t)(pl.dst - adam.grad);
    const float step_size = adam.hyper[0] / adam.hyper[1], rsqrt_bc2 = adam.hyper[2];
    const float mi = adam.b1 * adam.exp_avg[k] + (1.0f - adam.b1) * s;
    const float vi = adam.b2 * adam.exp_avg_sq[k] + (1.0f - adam.b2) * s * s;
    adam.exp_avg[k] = mi;
    adam.exp_avg_sq[k] = vi;
    const float p = adam.param[k] - step_size * (mi / (sqrtf(vi) * rsqrt_bc2 + adam.eps));
    adam.param[k] = p;
    if (adam.image) {   // next step's operand image without a prepare pass (its padding entries never change)
        adam.image[pl.pos_f] = (_Float16)p;
        adam.image[pl.pos_t] = (_Float16)p;
    }
}

// One group = 64 outputs, reduced by 256 lanes (tid 0..255; every lane of the group must call: there is a barrier inside).
// Wave q sums the slabs q, q + 4, q + 8, ... (eight loads in flight: the sum is latency-bound otherwise) and the four partial
// sums are added in wave order -- a fixed order, so the result is reproducible run to run.
// A kernel of its own (ngp_x_mlp_reduce_dw), or extra workgroups of another launch: the reduction is ten microseconds of
// latency-bound loads that nothing waits for until the next step's MLP forward.
__device__ __forceinline__ void mlp_reduce_dw_group(const MlpDwReduce &a, uint32_t group, uint32_t tid, float (*part)[64])
{
    const uint32_t e = group * 64 + (tid & 63u), q = tid >> 6;
    const bool view = e >= kAccFloats;
    const uint32_t i = view ? e - kAccFloats : e;
    const float *src = (view ? a.part_view : a.part_grid) + i;
    float s = 0.0f;
    uint32_t w = q;
    for (; w + 28 < a.n_wg; w += 32) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = src[(size_t)(w + 4 * k) * kAccFloats];
#pragma unroll
        for (int k = 0; k < 8; k++) s += t[k];
    }
    for (; w < a.n_wg; w += 4) s += src[(size_t)w * kAccFloats];
    part[q][tid & 63u] = s;
    __syncthreads();
    if (q != 0) return;
    s = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
    s *= a.scaler ? a.scaler[LS_INV] : a.inv_loss_scale;
    const DwPlace pl = mlp_dw_place(a, e);
    if (!pl.dst) return;   // padding elements of the tiles have no weight behind them
    *pl.dst = s;
    const bool finite = fabsf(s) < __uint_as_float(0x7f800000u);
    if (a.scaler) {
        // dynamic loss scale: an f16 delta that overflowed has made the weight gradients of its layer non-finite (inf x
        // activation, or NaN out of the on-matrix-core transposes) -- this is where the step's overflow is detected.  Whether
        // the optimiser steps is decided for the WHOLE step once every gradient is known: Adam on the MLP weights follows in
        // a later launch (mlp_adam_group: passengers of the table's reduce kernel, or ngp_x_adam_step_dev with the flag).
        if (!finite) reinterpret_cast<uint32_t *>(a.scaler)[LS_FOUND] = 1u;
        return;
    }
    // static loss scale: (a non-finite weight gradient -- one NaN dsigma / drgb in the batch -- leaves this weight, its
    // moments and its image entries alone: the update would poison it for good.  The table's half of that rule is in
    // bin_reduce_kernel, which skips the whole table when the batch's largest feature gradient is not finite.  The skip is
    // per MLP weight, not per step: this reduction runs beside the fill, before the batch's maximum is known; the step
    // counter and the bias corrections advance either way.)
    if (a.adam.param && finite) mlp_adam_element(a.adam, pl, s);
}

// Dynamic loss scale: Adam on the MLP weights (+ operand image) from the gradients a previous launch's mlp_reduce_dw_group
// left in a.dw*, element e of 2 kAccFloats; the caller has already decided that the step is not skipped.
__device__ __forceinline__ void mlp_adam_group(const MlpDwReduce &a, uint32_t e)
{
    const DwPlace pl = mlp_dw_place(a, e);
    if (pl.dst) mlp_adam_element(a.adam, pl, *pl.dst);
}

}  // namespace ngp
