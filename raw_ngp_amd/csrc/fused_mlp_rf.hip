// Fused tiny-MLP field of the light-conditioned configuration (`--rfield`, nerf/network.py:55-56 of the reference):
//     h     = W3 relu(W2 relu(W1 (enc * level_window)))                      32 -> 64 -> 64 -> 16   (grid_mlp)
//     sigma = exp(h[0])                                                      (trunc_exp, activation.py:9-19)
//     rgb   = min(exp(W6 relu(W5 relu(W4 [h[1:16], SH16(d), SH16(l)])) - 5), 5)   47 -> 80 -> 80 -> 3 (view_mlp)
// with the BARF level window of network.py:99-109 folded into the encoder read (f * w) and, in the backward, into the
// encoder gradient.  Same design as fused_mlp.hip / fused_mlp_backward.hip (one wave = 32 samples, layers chained through
// v_mfma_f32_32x32x16_f16 accumulators, f16 operands, f32 accumulation); the 80-wide layers are three 32-row blocks
// with rows 80..95 structurally zero, i.e. FIVE k-steps of 16 (the sixth is skipped), the 47 inputs are three k-steps:
//     k-step 0 = [sigma row (zero weight), 15 geometry features]   k-step 1 = SH16(view dir)   k-step 2 = SH16(light dir)
// The backward also returns what pose refinement needs (raymarching.py:319-329 sums it per ray): the gradient with
// respect to the UN-normalised view direction, through the SH basis' ambient Jacobian (shencoder.cu:126-350) and the
// normalisation d / |d| (renderer.py:541, sphere_harmonics.py:78-81).  Light directions are data: no gradient.
//
// Weight gradients: 18 accumulator tiles (dW4 6, dW5 9, dW6 3) do not fit next to the chain in one wave, so the view
// half is two kernels:
//   v1   recompute everything, delta chain, dW5; writes delta3 (-> grid kernel), d dirs, and delta4 / h4 / delta6 operands
//   v2   recompute layers 1-3 (the view MLP's inputs), reads delta4 / h4 / delta6; dW4, dW6
//   grid the 64-wide density MLP's kernel of fused_mlp_backward.hip with the level window (dW1-3, d enc)
#include "mlp_common.hpp"
#include "sh_eval.hpp"

namespace ngp {

// fragment ids of the rfield operand image (each fragment = 64 lanes x 8 halfs = 1 KiB)
enum RfFrag : uint32_t {
    RF_W1 = 0,    // [rb 2][s 2]        as F_W1
    RF_W2 = 4,    // [rb 2][kb 2][s 2]  as F_W2
    RF_W3 = 12,   // [kb 2][s 2]        as F_W3
    RF_W4 = 16,   // [rb 3][ks 3]       row 32rb + r < 80, input k in 1..47 ? W4[row][k - 1] : 0
    RF_W5 = 25,   // [rb 3][kk 5]       W5[32rb + r][k], k = 16kk + ...
    RF_W6 = 40,   // [kk 5]             r < 3 ? W6[r][k] : 0
    RF_T6 = 45,   // [rb 3]   (k-step 0) k < 3 ? W6[k][32rb + r] : 0
    RF_T5 = 48,   // [rb 3][kk 5]       W5[k][32rb + r]
    RF_T4 = 63,   // [kk 5]   (row block 0: inputs 0..31)  r >= 1 ? W4[k][r - 1] : 0
    RF_T3 = 68,   // [rb 2]             as T_W3
    RF_T2 = 70,   // [rb 2][kb 2][s 2]  as T_W2
    RF_T1 = 78,   // [kb 2][s 2]        as T_W1
    kRfFrags = 82,
};
constexpr uint32_t kRfImageHalfs = kRfFrags * 64 * 8;
constexpr uint32_t kRfHid = 80, kRfIn = 47;

__device__ __forceinline__ float rf_frag_elem(uint32_t f, uint32_t r, uint32_t h, uint32_t t, const MlpWeights &W)
{
    if (f < RF_W4) return frag_elem(f, r, h, t, W);                           // density MLP, forward blocks
    if (f >= RF_T3) return frag_elem(f - RF_T3 + T_W3, r, h, t, W);            // density MLP, transposed blocks
    if (f < RF_W5) {
        const uint32_t i = f - RF_W4, rb = i / 3, ks = i - rb * 3, k = kperm(ks, h, t), o = 32 * rb + r;
        return (o < kRfHid && k >= 1 && k <= kRfIn) ? W.w4[o * kRfIn + k - 1] : 0.0f;
    }
    if (f < RF_W6) {
        const uint32_t i = f - RF_W5, rb = i / 5, kk = i - rb * 5, k = kperm(kk, h, t), o = 32 * rb + r;
        return o < kRfHid ? W.w5[o * kRfHid + k] : 0.0f;
    }
    if (f < RF_T6) {
        const uint32_t kk = f - RF_W6;
        return r < 3 ? W.w6[r * kRfHid + kperm(kk, h, t)] : 0.0f;
    }
    if (f < RF_T5) {
        const uint32_t rb = f - RF_T6, k = kperm(0, h, t), o = 32 * rb + r;
        return (k < 3 && o < kRfHid) ? W.w6[k * kRfHid + o] : 0.0f;
    }
    if (f < RF_T4) {
        const uint32_t i = f - RF_T5, rb = i / 5, kk = i - rb * 5, o = 32 * rb + r;
        return o < kRfHid ? W.w5[kperm(kk, h, t) * kRfHid + o] : 0.0f;
    }
    const uint32_t kk = f - RF_T4;
    return r >= 1 ? W.w4[kperm(kk, h, t) * kRfIn + r - 1] : 0.0f;
}

__global__ __launch_bounds__(256) void mlp_rf_prepare_kernel(MlpWeights W, _Float16 *__restrict__ image)
{
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kRfImageHalfs) return;
    const uint32_t t = e & 7u, lane = (e >> 3) & 63u, f = e >> 9;
    image[e] = (_Float16)rf_frag_elem(f, lane & 31u, lane >> 5, t, W);
}

#define RF_FRAG(id) lds_w[(id) * 64 + lane]

template <int S>
__device__ __forceinline__ half8 pack_masked_sat(const f32x16 &a, const half8 &act, _Float16 lim)
{
    typedef short short8 __attribute__((ext_vector_type(8)));
    const half8 o = pack_sat<S>(a, lim);
    short8 m = __builtin_bit_cast(short8, act);
    m = __builtin_elementwise_min(__builtin_elementwise_max(m, (short8)0), (short8)1) * (short8)-1;
    return __builtin_bit_cast(half8, (short8)(__builtin_bit_cast(short8, o) & m));
}

__device__ __forceinline__ half8 rf_identity(uint32_t s, uint32_t lane)
{
    const uint32_t j = lane & 31u, h = lane >> 5;
    half8 o;
#pragma unroll
    for (uint32_t t = 0; t < 8; t++) o[t] = (_Float16)(kperm(s, h, t) == j ? 1.0f : 0.0f);
    return o;
}

// encoder features of sample `row` as the two B fragments of layer 1 (f32 window product first, like f * weights in
// network.py:109 under autocast, then f16)
__device__ __forceinline__ void load_enc(const float *__restrict__ enc, size_t stride, uint32_t row, bool valid, uint32_t h,
                                         const LaneWindow &lw, half8 (&x0)[2])
{
#pragma unroll
    for (uint32_t s = 0; s < 2; s++)
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            const uint32_t level = 8 * s + 4 * q + 2 * h;
            float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
            if (valid) {
                a = reinterpret_cast<const float2 *>(enc)[(size_t)level * stride + row];
                b = reinterpret_cast<const float2 *>(enc)[(size_t)(level + 1) * stride + row];
            }
            const float wa = lw.w[4 * s + 2 * q], wb = lw.w[4 * s + 2 * q + 1];
            x0[s][4 * q + 0] = (_Float16)(a.x * wa);
            x0[s][4 * q + 1] = (_Float16)(a.y * wa);
            x0[s][4 * q + 2] = (_Float16)(b.x * wb);
            x0[s][4 * q + 3] = (_Float16)(b.y * wb);
        }
}

// SH16 of a direction as one k-step fragment: element t = SH index 8 (t >> 2) + 4 h + (t & 3)
__device__ __forceinline__ half8 sh_fragment(const float *__restrict__ v, uint32_t row, bool valid, uint32_t h)
{
    float dx = 0.f, dy = 0.f, dz = 1.f;
    if (valid) {
        dx = v[(size_t)row * 3];
        dy = v[(size_t)row * 3 + 1];
        dz = v[(size_t)row * 3 + 2];
    }
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    float sh[16], j0[1], j1[1], j2[1];
    sh_eval<4, false>(dx * inv, dy * inv, dz * inv, sh, j0, j1, j2);
    half8 o;
#pragma unroll
    for (uint32_t t = 0; t < 8; t++) {
        const float lo = sh[8 * (t >> 2) + (t & 3)], hi = sh[8 * (t >> 2) + 4 + (t & 3)];
        o[t] = (_Float16)(h ? hi : lo);
    }
    return o;
}

// layers 1-3 (density MLP) from the weight fragments in LDS: o = 16 output rows (row 0 = raw density)
__device__ __forceinline__ f32x16 density_mlp(const half8 *lds_w, uint32_t lane, const half8 (&x0)[2])
{
    f32x16 a[2];
    half8 x[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        a[rb] = zero16();
#pragma unroll
        for (int s = 0; s < 2; s++) a[rb] = mfma(RF_FRAG(RF_W1 + rb * 2 + s), x0[s], a[rb]);
    }
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
        x[kb][0] = pack<0, true>(a[kb]);
        x[kb][1] = pack<1, true>(a[kb]);
    }
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        a[rb] = zero16();
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int s = 0; s < 2; s++) a[rb] = mfma(RF_FRAG(RF_W2 + rb * 4 + kb * 2 + s), x[kb][s], a[rb]);
    }
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
        x[kb][0] = pack<0, true>(a[kb]);
        x[kb][1] = pack<1, true>(a[kb]);
    }
    f32x16 o = zero16();
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int s = 0; s < 2; s++) o = mfma(RF_FRAG(RF_W3 + kb * 2 + s), x[kb][s], o);
    return o;
}

// post-ReLU activations of an 80-wide layer as its five k-step fragments
__device__ __forceinline__ void pack_hidden(const f32x16 (&a)[3], half8 (&x)[5])
{
    x[0] = pack<0, true>(a[0]);
    x[1] = pack<1, true>(a[0]);
    x[2] = pack<0, true>(a[1]);
    x[3] = pack<1, true>(a[1]);
    x[4] = pack<0, true>(a[2]);
}

// the view MLP forward: h3, h4 (five fragments each) and the 3 raw colour rows (registers 0..2 of the h = 0 lanes)
__device__ __forceinline__ f32x16 view_mlp(const half8 *lds_w, uint32_t lane, const half8 &x3a, const half8 &shd,
                                           const half8 &shl, half8 (&h3)[5], half8 (&h4)[5])
{
    f32x16 a[3];
#pragma unroll
    for (int rb = 0; rb < 3; rb++) {
        a[rb] = zero16();
        a[rb] = mfma(RF_FRAG(RF_W4 + rb * 3 + 0), x3a, a[rb]);
        a[rb] = mfma(RF_FRAG(RF_W4 + rb * 3 + 1), shd, a[rb]);
        a[rb] = mfma(RF_FRAG(RF_W4 + rb * 3 + 2), shl, a[rb]);
    }
    pack_hidden(a, h3);
#pragma unroll
    for (int rb = 0; rb < 3; rb++) {
        a[rb] = zero16();
#pragma unroll
        for (int kk = 0; kk < 5; kk++) a[rb] = mfma(RF_FRAG(RF_W5 + rb * 5 + kk), h3[kk], a[rb]);
    }
    pack_hidden(a, h4);
    f32x16 c = zero16();
#pragma unroll
    for (int kk = 0; kk < 5; kk++) c = mfma(RF_FRAG(RF_W6 + kk), h4[kk], c);
    return c;
}

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 2) void mlp_rf_forward_kernel(const float *__restrict__ enc, uint32_t stride,
                                                               const float *__restrict__ dirs,
                                                               const float *__restrict__ ldirs,
                                                               const float *__restrict__ level_w,
                                                               const int32_t *__restrict__ M_dev, uint32_t M_host,
                                                               const half8 *__restrict__ image,
                                                               float *__restrict__ sigma, float *__restrict__ rgb,
                                                               FieldAct act = FieldAct{})
{
    extern __shared__ half8 lds_w[];   // fragments 0..44: every forward block
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
    const uint32_t M = M_dev ? min((uint32_t)max(M_dev[0], 0), M_host) : M_host;
    const uint32_t n_tiles = (M + 31u) >> 5;
    for (uint32_t i = threadIdx.x; i < (uint32_t)RF_T6 * 64u; i += 256) lds_w[i] = image[i];
    __syncthreads();
    const LaneWindow lw = load_window(level_w, h);

    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        asm volatile("" ::: "memory");   // the weight fragments stay in LDS (no hoisting of 45 KiB into VGPRs)
        const uint32_t row = tile * 32u + n;
        const bool valid = row < M;
        half8 x0[2];
        load_enc(enc, stride, row, valid, h, lw, x0);
        const f32x16 o = density_mlp(lds_w, lane, x0);
        const float sigma_raw = o[0];
        if (rgb == nullptr) {   // density-only query (density-grid refresh)
            if (valid && h == 0) sigma[row] = act_sigma(sigma_raw, act);
            continue;
        }
        const half8 x3a = pack<0, false>(o);
        const half8 shd = sh_fragment(dirs, row, valid, h), shl = sh_fragment(ldirs, row, valid, h);
        half8 h3[5], h4[5];
        const f32x16 c = view_mlp(lds_w, lane, x3a, shd, shl, h3, h4);
        if (valid && h == 0) {
            // (the field's output activations, network.py:115,131-135; the defaults are trunc_exp and clamp(exp(x - 5), max 5))
            sigma[row] = act_sigma(sigma_raw, act);
            rgb[(size_t)row * 3 + 0] = act_color(c[0], act);
            rgb[(size_t)row * 3 + 1] = act_color(c[1], act);
            rgb[(size_t)row * 3 + 2] = act_color(c[2], act);
        }
    }
}

// ------------------------------------------------------------------ backward helpers
// [32 features x 32 samples] held as one or two operand fragments -> [32 samples x 32 features]
__device__ __forceinline__ f32x16 transpose2(half8 f0, half8 f1, half8 I0, half8 I1)
{
    f32x16 d = zero16();
    d = mfma(f0, I0, d);
    d = mfma(f1, I1, d);
    return d;
}
__device__ __forceinline__ f32x16 transpose1(half8 f0, half8 I0) { return mfma(f0, I0, zero16()); }

constexpr uint32_t kRfScratch = 11;   // per lane: delta4 (5), h4 (5), delta6 (1) fragments, layout [fragment][2 row + h]
constexpr uint32_t kRfTilesV = 9, kRfTilesG = 8;

// ------------------------------------------------------------------ backward, view half 1: delta chain + dW5
__global__ __launch_bounds__(256, 1) void mlp_rf_backward_v1_kernel(
    const float *__restrict__ enc, uint32_t stride, const float *__restrict__ dirs, const float *__restrict__ ldirs,
    const float *__restrict__ level_w, const float *__restrict__ dsigma, const float *__restrict__ drgb,
    const int32_t *__restrict__ M_dev, uint32_t M_host, const half8 *__restrict__ image, float loss_scale_host,
    half8 *__restrict__ d3buf, half8 *__restrict__ scratch, float *__restrict__ ddirs, float *__restrict__ partial,
    const int32_t *__restrict__ live_idx, const float *__restrict__ scaler, FieldAct act = FieldAct{})
{
    extern __shared__ half8 lds_w[];   // fragments 0..67 (68 KiB); reused as the f32 reduction image at the end
    // (scaler: the dynamic loss scale, as in fused_mlp_backward.hip's view kernel)
    const float loss_scale = scaler ? scaler[LS_SCALE] : loss_scale_host;
    const _Float16 lim = delta_limit(scaler);
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
    const uint32_t M = M_dev ? min((uint32_t)max(M_dev[0], 0), M_host) : M_host;
    const uint32_t n_tiles = (M + 31u) >> 5;
    const size_t plane = (size_t)M_host * 2;   // fragments of the scratch area are planes of 2 * M_host half8
    for (uint32_t i = threadIdx.x; i < (uint32_t)RF_T3 * 64u; i += 256) lds_w[i] = image[i];
    __syncthreads();
    const half8 I0 = rf_identity(0, lane), I1 = rf_identity(1, lane);
    const LaneWindow lw = load_window(level_w, h);
    const float inv_scale = 1.0f / loss_scale;

    f32x16 g[kRfTilesV];   // dW5[rb][cb] at rb * 3 + cb
#pragma unroll
    for (int i = 0; i < (int)kRfTilesV; i++) g[i] = zero16();

    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        asm volatile("" ::: "memory");
        // live_idx: the kernels run over a LIST of M samples (fused_mlp_backward.hip: the view kernel); inputs and d dirs by
        // sample (`row`), delta3 and the operands of the second view kernel in list order (`c_idx`)
        const uint32_t c_idx = tile * 32u + n;
        const bool valid = c_idx < M;
        const uint32_t row = (live_idx && valid) ? (uint32_t)live_idx[c_idx] : c_idx;
        float gs = 0.f, gr0 = 0.f, gr1 = 0.f, gr2 = 0.f;
        if (valid) {
            gs = dsigma[row];
            gr0 = drgb[(size_t)row * 3];
            gr1 = drgb[(size_t)row * 3 + 1];
            gr2 = drgb[(size_t)row * 3 + 2];
        }
        // samples behind the compositor's early stop have exactly zero output gradients: a tile made of such samples
        // contributes nothing anywhere (v2 and the grid kernel test the same thing and skip it too)
        if (__ballot(gs != 0.0f || gr0 != 0.0f || gr1 != 0.0f || gr2 != 0.0f) == 0ull) {
            if (valid) {
                half8 z;
#pragma unroll
                for (int t = 0; t < 8; t++) z[t] = (_Float16)0.0f;
                d3buf[(size_t)c_idx * 2 + h] = z;
                if (ddirs && h == 0) ddirs[(size_t)row * 3] = ddirs[(size_t)row * 3 + 1] = ddirs[(size_t)row * 3 + 2] = 0.0f;
            }
            continue;
        }
        // ---------------- recompute the forward pass
        half8 x0[2];
        load_enc(enc, stride, row, valid, h, lw, x0);
        const f32x16 o = density_mlp(lds_w, lane, x0);
        const float sigma_raw = o[0];
        const half8 x3a = pack<0, false>(o);
        const half8 shd = sh_fragment(dirs, row, valid, h), shl = sh_fragment(ldirs, row, valid, h);
        half8 h3[5], h4[5];
        const f32x16 c = view_mlp(lds_w, lane, x3a, shd, shl, h3, h4);

        // ---------------- output deltas (scaled so that they survive f16)
        f32x16 d6 = zero16();
        if (h == 0) {   // d rgb / d raw (default: exp(raw - 5) where the clamp at 5 is inactive, else 0)
            const float e0 = act_dcolor(c[0], act), e1 = act_dcolor(c[1], act), e2 = act_dcolor(c[2], act);
            d6[0] = e0 != 0.0f ? gr0 * e0 * loss_scale : 0.0f;
            d6[1] = e1 != 0.0f ? gr1 * e1 * loss_scale : 0.0f;
            d6[2] = e2 != 0.0f ? gr2 * e2 * loss_scale : 0.0f;
        }
        const half8 p6 = pack_sat<0>(d6, lim);

        // ---------------- delta5 = W6^T delta6 (masked)
        half8 p5[5];
#pragma unroll
        for (int rb = 0; rb < 3; rb++) {
            const f32x16 dh = mfma(RF_FRAG(RF_T6 + rb), p6, zero16());
            p5[2 * rb] = pack_masked_sat<0>(dh, h4[2 * rb], lim);
            if (rb < 2) p5[2 * rb + 1] = pack_masked_sat<1>(dh, h4[2 * rb + 1], lim);
        }
        // ---------------- dW5 += delta5 x H3^T
        {
            half8 a5[3][2];
#pragma unroll
            for (int rb = 0; rb < 3; rb++) {
                const f32x16 tt = rb < 2 ? transpose2(p5[2 * rb], p5[rb < 2 ? 2 * rb + 1 : 0], I0, I1) : transpose1(p5[4], I0);
                a5[rb][0] = pack<0, false>(tt);
                a5[rb][1] = pack<1, false>(tt);
            }
#pragma unroll
            for (int cb = 0; cb < 3; cb++) {
                const f32x16 tt = cb < 2 ? transpose2(h3[2 * cb], h3[cb < 2 ? 2 * cb + 1 : 0], I0, I1) : transpose1(h3[4], I0);
                const half8 b0 = pack<0, false>(tt), b1 = pack<1, false>(tt);
#pragma unroll
                for (int rb = 0; rb < 3; rb++) {
                    g[rb * 3 + cb] = mfma(a5[rb][0], b0, g[rb * 3 + cb]);
                    g[rb * 3 + cb] = mfma(a5[rb][1], b1, g[rb * 3 + cb]);
                }
            }
        }
        // ---------------- delta4 = W5^T delta5 (masked)
        half8 p4[5];
#pragma unroll
        for (int rb = 0; rb < 3; rb++) {
            f32x16 dh = zero16();
#pragma unroll
            for (int kk = 0; kk < 5; kk++) dh = mfma(RF_FRAG(RF_T5 + rb * 5 + kk), p5[kk], dh);
            p4[2 * rb] = pack_masked_sat<0>(dh, h3[2 * rb], lim);
            if (rb < 2) p4[2 * rb + 1] = pack_masked_sat<1>(dh, h3[2 * rb + 1], lim);
        }
        // ---------------- d (view-MLP inputs 0..31) = W4^T delta4: row 0 unused, 1..15 features, 16..31 SH(view dir)
        f32x16 dx3 = zero16();
#pragma unroll
        for (int kk = 0; kk < 5; kk++) dx3 = mfma(RF_FRAG(RF_T4 + kk), p4[kk], dx3);
        // delta3 row 0 = d sigma_raw = dsigma * d sigma / d raw   (default: exp(clamp(raw, -80, 80)), trunc_exp's backward)
        if (h == 0) dx3[0] = gs * act_dsigma(sigma_raw, act) * loss_scale;
        if (valid) d3buf[(size_t)c_idx * 2 + h] = pack_sat<0>(dx3, lim);

        if (ddirs) {   // d loss / d (un-normalised view direction)
            float dx = 0.f, dy = 0.f, dz = 1.f;
            if (valid) {
                dx = dirs[(size_t)row * 3];
                dy = dirs[(size_t)row * 3 + 1];
                dz = dirs[(size_t)row * 3 + 2];
            }
            const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
            const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
            float sh[16], jx[16], jy[16], jz[16];
            sh_eval<4, true>(ux, uy, uz, sh, jx, jy, jz);
            float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
            for (uint32_t t = 0; t < 8; t++) {   // register 8 + t = SH index 8 (t >> 2) + 4 h + (t & 3)
                const uint32_t lo = 8 * (t >> 2) + (t & 3), hi = lo + 4;
                const float gv = dx3[8 + t];
                gx = fmaf(gv, h ? jx[hi] : jx[lo], gx);
                gy = fmaf(gv, h ? jy[hi] : jy[lo], gy);
                gz = fmaf(gv, h ? jz[hi] : jz[lo], gz);
            }
            gx += __shfl_xor(gx, 32, 64);
            gy += __shfl_xor(gy, 32, 64);
            gz += __shfl_xor(gz, 32, 64);
            // through u = d / |d| (applied twice by the reference -- renderer.py:541 and the encoder module -- the second
            // one on a unit vector: both Jacobians are the same tangent projection, which is idempotent)
            const float dot = gx * ux + gy * uy + gz * uz;
            const float k = inv * inv_scale;
            if (valid && h == 0) {
                ddirs[(size_t)row * 3] = (gx - ux * dot) * k;
                ddirs[(size_t)row * 3 + 1] = (gy - uy * dot) * k;
                ddirs[(size_t)row * 3 + 2] = (gz - uz * dot) * k;
            }
        }
        if (valid) {   // operands of the second view kernel
            const size_t at = (size_t)c_idx * 2 + h;
#pragma unroll
            for (int kk = 0; kk < 5; kk++) {
                scratch[(size_t)kk * plane + at] = p4[kk];
                scratch[(size_t)(5 + kk) * plane + at] = h4[kk];
            }
            scratch[(size_t)10 * plane + at] = p6;
        }
    }
    flush_tiles_parallel<(int)kRfTilesV>(reinterpret_cast<float *>(lds_w), g, lane, partial + (size_t)blockIdx.x * kRfTilesV * 1024u);
}

// ------------------------------------------------------------------ backward, view half 2: dW4, dW6
__global__ __launch_bounds__(256, 1) void mlp_rf_backward_v2_kernel(
    const float *__restrict__ enc, uint32_t stride, const float *__restrict__ dirs, const float *__restrict__ ldirs,
    const float *__restrict__ level_w, const float *__restrict__ dsigma, const float *__restrict__ drgb,
    const int32_t *__restrict__ M_dev, uint32_t M_host, const half8 *__restrict__ image,
    const half8 *__restrict__ scratch, float *__restrict__ partial, const int32_t *__restrict__ live_idx)
{
    extern __shared__ half8 lds_w[];   // fragments 0..15 (density MLP forward); sized for the 36 KiB reduction image
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
    const uint32_t M = M_dev ? min((uint32_t)max(M_dev[0], 0), M_host) : M_host;
    const uint32_t n_tiles = (M + 31u) >> 5;
    const size_t plane = (size_t)M_host * 2;
    for (uint32_t i = threadIdx.x; i < (uint32_t)RF_W4 * 64u; i += 256) lds_w[i] = image[i];
    __syncthreads();
    const half8 I0 = rf_identity(0, lane), I1 = rf_identity(1, lane);
    const LaneWindow lw = load_window(level_w, h);

    f32x16 g[kRfTilesV];   // 0..5: dW4[rb][cb] at rb * 2 + cb   6..8: dW6[cb]
#pragma unroll
    for (int i = 0; i < (int)kRfTilesV; i++) g[i] = zero16();

    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        asm volatile("" ::: "memory");
        const uint32_t c_idx = tile * 32u + n;   // (list order; `row`: the sample -- see v1)
        const bool valid = c_idx < M;
        const uint32_t row = (live_idx && valid) ? (uint32_t)live_idx[c_idx] : c_idx;
        float gs = 0.f, gr0 = 0.f, gr1 = 0.f, gr2 = 0.f;
        if (valid) {
            gs = dsigma[row];
            gr0 = drgb[(size_t)row * 3];
            gr1 = drgb[(size_t)row * 3 + 1];
            gr2 = drgb[(size_t)row * 3 + 2];
        }
        if (__ballot(gs != 0.0f || gr0 != 0.0f || gr1 != 0.0f || gr2 != 0.0f) == 0ull) continue;   // as v1
        half8 p4[5], h4[5], p6;
        {
            half8 z;
#pragma unroll
            for (int t = 0; t < 8; t++) z[t] = (_Float16)0.0f;
            const size_t at = (size_t)c_idx * 2 + h;
#pragma unroll
            for (int kk = 0; kk < 5; kk++) {
                p4[kk] = valid ? scratch[(size_t)kk * plane + at] : z;
                h4[kk] = valid ? scratch[(size_t)(5 + kk) * plane + at] : z;
            }
            p6 = valid ? scratch[(size_t)10 * plane + at] : z;
        }
        half8 x0[2];
        load_enc(enc, stride, row, valid, h, lw, x0);
        const f32x16 o = density_mlp(lds_w, lane, x0);
        const half8 x3a = pack<0, false>(o);
        const half8 shd = sh_fragment(dirs, row, valid, h), shl = sh_fragment(ldirs, row, valid, h);

        // ---------------- dW4 += delta4 x X3^T   (inputs 0..31 = [sigma row, features, SH(d)], 32..47 = SH(l))
        {
            const f32x16 t0 = transpose2(x3a, shd, I0, I1), t1 = transpose1(shl, I0);
            const half8 b00 = pack<0, false>(t0), b01 = pack<1, false>(t0), b10 = pack<0, false>(t1), b11 = pack<1, false>(t1);
#pragma unroll
            for (int rb = 0; rb < 3; rb++) {
                const f32x16 tt = rb < 2 ? transpose2(p4[2 * rb], p4[rb < 2 ? 2 * rb + 1 : 0], I0, I1) : transpose1(p4[4], I0);
                const half8 a0 = pack<0, false>(tt), a1 = pack<1, false>(tt);
                g[rb * 2 + 0] = mfma(a0, b00, g[rb * 2 + 0]);
                g[rb * 2 + 0] = mfma(a1, b01, g[rb * 2 + 0]);
                g[rb * 2 + 1] = mfma(a0, b10, g[rb * 2 + 1]);
                g[rb * 2 + 1] = mfma(a1, b11, g[rb * 2 + 1]);
            }
        }
        // ---------------- dW6 += delta6 x H4^T
        {
            const f32x16 t6 = transpose1(p6, I0);
            const half8 a0 = pack<0, false>(t6), a1 = pack<1, false>(t6);
#pragma unroll
            for (int cb = 0; cb < 3; cb++) {
                const f32x16 tt = cb < 2 ? transpose2(h4[2 * cb], h4[cb < 2 ? 2 * cb + 1 : 0], I0, I1) : transpose1(h4[4], I0);
                const half8 b0 = pack<0, false>(tt), b1 = pack<1, false>(tt);
                g[6 + cb] = mfma(a0, b0, g[6 + cb]);
                g[6 + cb] = mfma(a1, b1, g[6 + cb]);
            }
        }
    }
    flush_tiles_parallel<(int)kRfTilesV>(reinterpret_cast<float *>(lds_w), g, lane, partial + (size_t)blockIdx.x * kRfTilesV * 1024u);
}

// ------------------------------------------------------------------ partial-slab reduction
// element e of the three slabs of a workgroup: [v1: 9 tiles | v2: 9 tiles | grid: 8 tiles] x 1024 floats;
// inside a tile: register v = (e / 64) % 16, lane = e % 64 -> row o = (v&3) + 8(v>>2) + 4(lane>>5), column j = lane & 31
struct RfGrads {
    float *dw1, *dw2, *dw3, *dw4, *dw5, *dw6;
};
__global__ __launch_bounds__(256) void mlp_rf_reduce_dw_kernel(const float *__restrict__ part_v1,
                                                              const float *__restrict__ part_v2,
                                                              const float *__restrict__ part_g, uint32_t n_wg,
                                                              float inv_loss_scale, RfGrads G, float *__restrict__ scaler)
{
    __shared__ float part[4][64];
    const uint32_t e = blockIdx.x * 64 + (threadIdx.x & 63u), q = threadIdx.x >> 6;
    constexpr uint32_t nV = kRfTilesV * 1024u, nG = kRfTilesG * 1024u;
    const uint32_t which = e < nV ? 0u : (e < 2 * nV ? 1u : 2u);
    const uint32_t i = which == 0 ? e : (which == 1 ? e - nV : e - 2 * nV);
    const uint32_t slab = which == 2 ? nG : nV;
    const float *src = (which == 0 ? part_v1 : (which == 1 ? part_v2 : part_g)) + i;
    float s = 0.0f;
    uint32_t w = q;
    for (; w + 28 < n_wg; w += 32) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = src[(size_t)(w + 4 * k) * slab];
#pragma unroll
        for (int k = 0; k < 8; k++) s += t[k];
    }
    for (; w < n_wg; w += 4) s += src[(size_t)w * slab];
    part[q][threadIdx.x & 63u] = s;
    __syncthreads();
    if (q != 0) return;
    s = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
    s *= scaler ? scaler[LS_INV] : inv_loss_scale;
    const uint32_t b = i >> 10, v = (i >> 6) & 15u, lane = i & 63u;
    const uint32_t o = (v & 3u) + 8u * (v >> 2) + 4u * (lane >> 5), j = lane & 31u;
    float *dst = nullptr;
    if (which == 0) {   // dW5[rb][cb]
        const uint32_t rb = b / 3, cb = b - rb * 3, r = 32 * rb + o, c = 32 * cb + j;
        if (r < kRfHid && c < kRfHid) dst = &G.dw5[r * kRfHid + c];
    } else if (which == 1) {
        if (b < 6) {   // dW4[rb][cb]: column = input index, 0 is the sigma row
            const uint32_t rb = b >> 1, cb = b & 1u, r = 32 * rb + o, k = 32 * cb + j;
            if (r < kRfHid && k >= 1 && k <= kRfIn) dst = &G.dw4[r * kRfIn + k - 1];
        } else {       // dW6[cb]
            const uint32_t c = 32 * (b - 6) + j;
            if (o < 3 && c < kRfHid) dst = &G.dw6[o * kRfHid + c];
        }
    } else {
        if (b < 2) {
            dst = &G.dw1[(32 * b + o) * 32 + j];
        } else if (b < 6) {
            const uint32_t rb = (b - 2) >> 1, cb = (b - 2) & 1;
            dst = &G.dw2[(32 * rb + o) * 64 + 32 * cb + j];
        } else if (o < 16) {
            dst = &G.dw3[o * 64 + 32 * (b - 6) + j];
        }
    }
    if (dst) *dst = s;
    // dynamic loss scale: an overflowed delta shows here (mlp_common.hpp: mlp_reduce_dw_group); the optimiser kernels skip
    if (dst && scaler && !(fabsf(s) < __uint_as_float(0x7f800000u))) reinterpret_cast<uint32_t *>(scaler)[LS_FOUND] = 1u;
}

}  // namespace ngp

using namespace ngp;

extern "C" size_t ngp_x_mlp_rf_image_bytes(void) { return (size_t)kRfImageHalfs * 2; }

extern "C" int ngp_x_mlp_rf_prepare(const float *w1, const float *w2, const float *w3, const float *w4, const float *w5,
                                    const float *w6, void *image, ngp_stream_t stream)
{
    NGP_REQUIRE(w1 && w2 && w3 && w4 && w5 && w6 && image, "mlp_rf_prepare: null tensor");
    NGP_REQUIRE(((uintptr_t)image & 15u) == 0, "mlp_rf_prepare: image must be 16-byte aligned");
    MlpWeights W{w1, w2, w3, w4, w5, w6};
    mlp_rf_prepare_kernel<<<dim3(ceil_div(kRfImageHalfs, 256u)), dim3(256), 0, as_stream(stream)>>>(
        W, reinterpret_cast<_Float16 *>(image));
    NGP_CHECK_LAUNCH("mlp_rf_prepare");
    return NGP_OK;
}

extern "C" int ngp_x_mlp_rf_forward(const float *enc, uint32_t stride, const float *dirs, const float *ldirs,
                                    const float *level_w, const int32_t *M_dev, uint32_t M, const void *image,
                                    float *sigma, float *rgb, ngp_stream_t stream)
{
    return ngp_x_mlp_rf_forward_act(enc, stride, dirs, ldirs, level_w, M_dev, M, image, sigma, rgb, 0, 0, 1.0f, stream);
}

// ... with the field's other OUTPUT activations (as ngp_x_mlp_forward_act: color_act 0 clamped_exp / 1 exp / 2 sigmoid,
// density_act 0 trunc_exp / 1 softplus(beta, threshold 20)); the hidden layers of this field are ReLU
extern "C" int ngp_x_mlp_rf_forward_act(const float *enc, uint32_t stride, const float *dirs, const float *ldirs,
                                        const float *level_w, const int32_t *M_dev, uint32_t M, const void *image,
                                        float *sigma, float *rgb, uint32_t color_act, uint32_t density_act, float beta,
                                        ngp_stream_t stream)
{
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(color_act <= 2u && density_act <= 1u && beta > 0.0f, "mlp_rf_forward: unknown activation or beta <= 0");
    FieldAct act;
    act.color = color_act, act.density = density_act, act.beta = beta, act.internal = 0;
    NGP_REQUIRE(enc && image && sigma, "mlp_rf_forward: null tensor");
    NGP_REQUIRE(rgb == nullptr || (dirs && ldirs), "mlp_rf_forward: dirs / ldirs missing");
    NGP_REQUIRE(stride >= M, "mlp_rf_forward: encoder slab stride smaller than M");
    const uint32_t tiles = ceil_div(M, 32u);
    const uint32_t blocks = min(ceil_div(tiles, 4u), 256u * 2u);
    mlp_rf_forward_kernel<<<dim3(blocks), dim3(256), (size_t)RF_T6 * 1024, as_stream(stream)>>>(
        enc, stride, dirs, ldirs, level_w, M_dev, M, reinterpret_cast<const half8 *>(image), sigma, rgb, act);
    NGP_CHECK_LAUNCH("mlp_rf_forward");
    return NGP_OK;
}

static uint32_t rf_bwd_blocks(uint32_t M)
{
    const uint32_t tiles = ceil_div(M, 32u);
    return max(1u, min(ceil_div(tiles, 4u), 256u));
}

// workspace: [delta3: M x 32 B][scratch: 11 planes of M x 32 B][partials v1 | v2 | grid, 256 workgroups each]
static size_t rf_off_scratch(uint32_t M) { return ((size_t)M * 32 + 255) & ~(size_t)255; }
static size_t rf_off_partial(uint32_t M) { return rf_off_scratch(M) + (((size_t)M * 32 * kRfScratch + 255) & ~(size_t)255); }

extern "C" size_t ngp_x_mlp_rf_backward_workspace_bytes(uint32_t M)
{
    return rf_off_partial(M) + (size_t)256 * (2 * kRfTilesV + kRfTilesG) * 1024 * 4 + 256;
}

extern "C" int ngp_x_mlp_rf_backward(const float *enc, uint32_t stride, const float *dirs, const float *ldirs,
                                     const float *level_w, const float *dsigma, const float *drgb, const int32_t *M_dev,
                                     uint32_t M, const void *image, float loss_scale, float *denc, float *ddirs,
                                     float *dw1, float *dw2, float *dw3, float *dw4, float *dw5, float *dw6,
                                     void *workspace, size_t workspace_bytes, ngp_stream_t stream)
{
    return ngp_x_mlp_rf_backward_list(enc, stride, dirs, ldirs, level_w, dsigma, drgb, M_dev, M, nullptr, image, loss_scale, denc,
                                      ddirs, dw1, dw2, dw3, dw4, dw5, dw6, workspace, workspace_bytes, nullptr, stream);
}

// ... over a LIST of samples (as ngp_x_mlp_backward_list): inputs and ddirs by sample, denc in list order
extern "C" int ngp_x_mlp_rf_backward_list(const float *enc, uint32_t stride, const float *dirs, const float *ldirs,
                                          const float *level_w, const float *dsigma, const float *drgb,
                                          const int32_t *M_dev, uint32_t M, const int32_t *sample_index, const void *image,
                                          float loss_scale, float *denc, float *ddirs, float *dw1, float *dw2, float *dw3,
                                          float *dw4, float *dw5, float *dw6, void *workspace, size_t workspace_bytes,
                                          float *loss_scaler, ngp_stream_t stream)
{
    return ngp_x_mlp_rf_backward_act(enc, stride, dirs, ldirs, level_w, dsigma, drgb, M_dev, M, sample_index, image, loss_scale,
                                     denc, ddirs, dw1, dw2, dw3, dw4, dw5, dw6, workspace, workspace_bytes, loss_scaler, 0, 0,
                                     1.0f, stream);
}

// ... with the output activations of ngp_x_mlp_rf_forward_act (their derivatives enter the output deltas)
extern "C" int ngp_x_mlp_rf_backward_act(const float *enc, uint32_t stride, const float *dirs, const float *ldirs,
                                         const float *level_w, const float *dsigma, const float *drgb,
                                         const int32_t *M_dev, uint32_t M, const int32_t *sample_index, const void *image,
                                         float loss_scale, float *denc, float *ddirs, float *dw1, float *dw2, float *dw3,
                                         float *dw4, float *dw5, float *dw6, void *workspace, size_t workspace_bytes,
                                         float *loss_scaler, uint32_t color_act, uint32_t density_act, float beta,
                                         ngp_stream_t stream)
{
    NGP_REQUIRE(color_act <= 2u && density_act <= 1u && beta > 0.0f, "mlp_rf_backward: unknown activation or beta <= 0");
    FieldAct act;
    act.color = color_act, act.density = density_act, act.beta = beta, act.internal = 0;
    NGP_REQUIRE(image && workspace && dw1 && dw2 && dw3 && dw4 && dw5 && dw6, "mlp_rf_backward: null tensor");
    NGP_REQUIRE(M == 0 || (enc && dirs && ldirs && dsigma && drgb && denc), "mlp_rf_backward: null sample tensor");
    NGP_REQUIRE(stride >= M, "mlp_rf_backward: encoder slab stride smaller than M");
    NGP_REQUIRE(workspace_bytes >= ngp_x_mlp_rf_backward_workspace_bytes(M), "mlp_rf_backward: workspace too small");
    NGP_REQUIRE(((uintptr_t)workspace & 15u) == 0, "mlp_rf_backward: workspace must be 16-byte aligned");
    NGP_REQUIRE(loss_scale > 0.0f, "mlp_rf_backward: loss_scale must be positive");
    constexpr size_t kV1Lds = flush_lds_bytes((int)kRfTilesV) > (size_t)RF_T3 * 1024 ? flush_lds_bytes((int)kRfTilesV) : (size_t)RF_T3 * 1024;
    constexpr size_t kV2Lds = flush_lds_bytes((int)kRfTilesV);
    static const bool lds_ok = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_rf_backward_v1_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kV1Lds) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_rf_backward_v2_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kV2Lds) == hipSuccess;
    }();
    NGP_REQUIRE(lds_ok, "mlp_rf_backward: cannot raise the dynamic LDS limit");
    hipStream_t st = as_stream(stream);
    const uint32_t blocks = rf_bwd_blocks(max(M, 1u));
    char *ws = reinterpret_cast<char *>(workspace);
    half8 *d3buf = reinterpret_cast<half8 *>(ws);
    half8 *scratch = reinterpret_cast<half8 *>(ws + rf_off_scratch(M));
    float *part_v1 = reinterpret_cast<float *>(ws + rf_off_partial(M));
    float *part_v2 = part_v1 + (size_t)256 * kRfTilesV * 1024;
    float *part_g = part_v2 + (size_t)256 * kRfTilesV * 1024;
    const half8 *img = reinterpret_cast<const half8 *>(image);
    mlp_rf_backward_v1_kernel<<<dim3(blocks), dim3(256), kV1Lds, st>>>(
        enc, stride, dirs, ldirs, level_w, dsigma, drgb, M_dev, M, img, loss_scale, d3buf, scratch, ddirs, part_v1, sample_index,
        loss_scaler, act);
    mlp_rf_backward_v2_kernel<<<dim3(blocks), dim3(256), kV2Lds, st>>>(
        enc, stride, dirs, ldirs, level_w, dsigma, drgb, M_dev, M, img, scratch, part_v2, sample_index);
    const int rc = launch_mlp_backward_grid(enc, stride, level_w, M_dev, M, img, RF_T3, 1.0f / loss_scale, d3buf, denc,
                                            part_g, blocks, st, sample_index, loss_scaler);
    if (rc != NGP_OK) return rc;
    mlp_rf_reduce_dw_kernel<<<dim3((2 * kRfTilesV + kRfTilesG) * 1024u / 64u), dim3(256), 0, st>>>(
        part_v1, part_v2, part_g, blocks, 1.0f / loss_scale, RfGrads{dw1, dw2, dw3, dw4, dw5, dw6}, loss_scaler);
    NGP_CHECK_LAUNCH("mlp_rf_backward");
    return NGP_OK;
}

// ngp_x_mlp_density_gradient for the light-conditioned field's operand image (the density network is the same, its
// transposed fragments sit at RF_T3)
extern "C" int ngp_x_mlp_rf_density_gradient(const float *enc, uint32_t stride, const float *level_w, const int32_t *M_dev,
                                             uint32_t M, const void *image, float *denc, ngp_stream_t stream)
{
    return launch_mlp_density_gradient("mlp_rf_density_gradient", enc, stride, level_w, M_dev, M, image, RF_T3, denc,
                                       as_stream(stream));
}
