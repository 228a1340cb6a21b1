// Backward of the fused tiny-MLP field: weight gradients and the gradient w.r.t. the encoder features.
//
// Replaces the autograd backward of the six nn.Linear layers + activations of nerf/network.py:74-143
// (reference: cuBLAS/hipBLASLt GEMMs over [M, 64] activations stored in HBM).  Nothing is stored by the
// forward pass: each kernel recomputes the activations of its 32-sample tile in registers, then runs the
// delta chain with transposed weight fragments (mlp_common.hpp).  Weight gradients need the contraction
// over SAMPLES, which live on the lanes of every tile; instead of a round trip through LDS the tiles are
// transposed on the matrix cores themselves (tile^T = tile-as-A x identity-as-B, exact in f16), after
// which  dW[o][c] += delta^T-as-A x act^T-as-B  is again a plain MFMA.  The MFMA work roughly doubles
// (146 instead of ~70 per 32 samples); at 32 cycles each that is still ~20 us for 2^18 samples.
//
// Two kernels because 16 dW accumulator tiles (256 VGPRs) do not fit next to the chain:
//   view kernel  recompute all six layers; deltas of the colour MLP; dW4, dW5, dW6; writes delta3 (the
//                gradient at the 16 outputs of the density MLP, f16, operand order) to a scratch slab
//   grid kernel  recompute layers 1-2; reads delta3; dW1, dW2, dW3; writes d(enc) in the [L, stride, 2] layout
// Per-workgroup partial dW tiles go to a slab; ngp_x_mlp_backward's last kernel sums them, removes the
// loss scale and emits the six fp32 gradients in torch layout.
#include "mlp_common.hpp"
#include "sh_eval.hpp"

namespace ngp {

// dynamic LDS: the weight fragments during the tile loop (46 / 26 KiB), then the staging area of the partial-sum flush
constexpr size_t kViewLds = flush_lds_bytes(8) > 46 * 1024 ? flush_lds_bytes(8) : 46 * 1024;
constexpr size_t kGridLds = flush_lds_bytes(8) > 32 * 1024 ? flush_lds_bytes(8) : 32 * 1024;

// ---- helpers -----------------------------------------------------------------------------------
template <bool WANT_SH>
struct TileInB {
    half8 x0[2];
    half8 sh;
};

template <bool WANT_SH>
__device__ __forceinline__ TileInB<WANT_SH> load_tile_b(const float *__restrict__ enc, size_t stride,
                                                       const float *__restrict__ dirs, uint32_t row, bool valid,
                                                       uint32_t h)
{
    TileInB<WANT_SH> in;
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
            in.x0[s][4 * q + 0] = (_Float16)a.x;
            in.x0[s][4 * q + 1] = (_Float16)a.y;
            in.x0[s][4 * q + 2] = (_Float16)b.x;
            in.x0[s][4 * q + 3] = (_Float16)b.y;
        }
    if constexpr (WANT_SH) {
        float dx = 0.f, dy = 0.f, dz = 1.f;
        if (valid) {
            dx = dirs[(size_t)row * 3];
            dy = dirs[(size_t)row * 3 + 1];
            dz = dirs[(size_t)row * 3 + 2];
        }
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        float sh[16], j0[1], j1[1], j2[1];
        sh_eval<4, false>(dx * inv, dy * inv, dz * inv, sh, j0, j1, j2);
#pragma unroll
        for (uint32_t t = 0; t < 8; t++) {
            const float lo = sh[8 * (t >> 2) + (t & 3)], hi = sh[8 * (t >> 2) + 4 + (t & 3)];
            in.sh[t] = (_Float16)(h ? hi : lo);
        }
    }
    return in;
}

// One wave per SIMD (417 VGPRs) means nothing hides a tile's load latency but the wave itself: the raw inputs of
// the NEXT tile are requested before the current tile's MFMA chain starts and converted when their turn comes.
struct RawTile {
    float2 e[8];        // encoder rows: levels 8s + 4q + 2h and +1, in load_tile_b order
    float d[3];         // direction (view kernel)
    float gs, gr[3];    // d loss / d sigma, d rgb (view kernel)
    half8 p3;           // delta of the density MLP's output layer (grid kernel)
};

template <bool VIEW>
__device__ __forceinline__ RawTile fetch_raw(const float *__restrict__ enc, size_t stride, const float *__restrict__ dirs,
                                             const float *__restrict__ dsigma, const float *__restrict__ drgb,
                                             const half8 *__restrict__ d3buf, uint32_t row, uint32_t c, bool valid,
                                             uint32_t h)
{
    // row: the sample (encoder slab, directions, output gradients); c: its place in the list the kernel runs over (delta3)
    RawTile r;
#pragma unroll
    for (uint32_t s = 0; s < 2; s++)
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            const uint32_t level = 8 * s + 4 * q + 2 * h;
            r.e[4 * s + 2 * q] = make_float2(0.f, 0.f);
            r.e[4 * s + 2 * q + 1] = make_float2(0.f, 0.f);
            if (valid) {
                r.e[4 * s + 2 * q] = reinterpret_cast<const float2 *>(enc)[(size_t)level * stride + row];
                r.e[4 * s + 2 * q + 1] = reinterpret_cast<const float2 *>(enc)[(size_t)(level + 1) * stride + row];
            }
        }
    r.d[0] = r.d[1] = 0.f;
    r.d[2] = 1.f;
    r.gs = r.gr[0] = r.gr[1] = r.gr[2] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; t++) r.p3[t] = (_Float16)0.0f;
    if (valid) {
        if constexpr (VIEW) {
            r.d[0] = dirs[(size_t)row * 3];
            r.d[1] = dirs[(size_t)row * 3 + 1];
            r.d[2] = dirs[(size_t)row * 3 + 2];
            r.gs = dsigma[row];
            r.gr[0] = drgb[(size_t)row * 3];
            r.gr[1] = drgb[(size_t)row * 3 + 1];
            r.gr[2] = drgb[(size_t)row * 3 + 2];
        } else if (d3buf) {
            r.p3 = d3buf[(size_t)c * 2 + h];
        }
    }
    return r;
}

template <bool WANT_SH>
__device__ __forceinline__ TileInB<WANT_SH> convert_raw(const RawTile &r, uint32_t h)
{
    TileInB<WANT_SH> in;
#pragma unroll
    for (uint32_t s = 0; s < 2; s++)
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            const float2 a = r.e[4 * s + 2 * q], b = r.e[4 * s + 2 * q + 1];
            in.x0[s][4 * q + 0] = (_Float16)a.x;
            in.x0[s][4 * q + 1] = (_Float16)a.y;
            in.x0[s][4 * q + 2] = (_Float16)b.x;
            in.x0[s][4 * q + 3] = (_Float16)b.y;
        }
    if constexpr (WANT_SH) {
        const float dx = r.d[0], dy = r.d[1], dz = r.d[2];
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        float sh[16], j0[1], j1[1], j2[1];
        sh_eval<4, false>(dx * inv, dy * inv, dz * inv, sh, j0, j1, j2);
#pragma unroll
        for (uint32_t t = 0; t < 8; t++) {
            const float lo = sh[8 * (t >> 2) + (t & 3)], hi = sh[8 * (t >> 2) + 4 + (t & 3)];
            in.sh[t] = (_Float16)(h ? hi : lo);
        }
    }
    return in;
}

__device__ __forceinline__ half8 identity_frag(uint32_t s, uint32_t lane)
{
    const uint32_t j = lane & 31u, h = lane >> 5;
    half8 o;
#pragma unroll
    for (uint32_t t = 0; t < 8; t++) o[t] = (_Float16)(kperm(s, h, t) == j ? 1.0f : 0.0f);
    return o;
}

// [32 features x 32 samples] (two operand fragments) -> [32 samples x 32 features]
__device__ __forceinline__ f32x16 transpose_tile(half8 f0, half8 f1, half8 I0, half8 I1)
{
    f32x16 d = zero16();
    d = mfma(f0, I0, d);
    d = mfma(f1, I1, d);
    return d;
}

#define NGP_FRAG(local_id) lds_w[(local_id) * 64 + lane]
// Diagnostic build only (-DNGP_STAMP, never shipped): s_memtime stamps around the sections of a tile, summed per wave and
// added to a device array that tools read back with ngp_dbg_read_stamps().  No stamp executes in the product build.
#ifdef NGP_STAMP
__device__ unsigned long long ngp_dbg_stamps[16];
#define NGP_STAMP_DECL unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = 0, st_t1 = 0
#define NGP_STAMP_BEGIN()                                                                         \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#define NGP_STAMP_AT(i)                                                                           \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t1)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        st_acc[i] += st_t1 - st_t0;                                                               \
        st_t0 = st_t1;                                                                            \
    } while (0)
#define NGP_STAMP_FLUSH()                                                                         \
    do {                                                                                          \
        if ((threadIdx.x & 63u) == 0)                                                             \
            for (int i_ = 0; i_ < 10; i_++) atomicAdd(&ngp_dbg_stamps[i_], st_acc[i_]);            \
    } while (0)
#else
#define NGP_STAMP_DECL
#define NGP_STAMP_BEGIN()
#define NGP_STAMP_AT(i)
#define NGP_STAMP_FLUSH()
#endif

// ---- view kernel ---------------------------------------------------------------------------------
// DDIRS: also d loss / d (un-normalised view direction) [M,3] -- pose refinement without the light-conditioned field: the
// gradient with respect to the SH features is rows 16..31 of d x3; the SH Jacobian and the tangent projection of d / |d|
// follow as in fused_mlp_rf.hip
template <bool DDIRS, bool SOFT = false>
__global__ __launch_bounds__(256, 1) void mlp_backward_view_kernel(
    const float *__restrict__ enc, uint32_t stride, const float *__restrict__ dirs, const float *__restrict__ dsigma,
    const float *__restrict__ drgb, const int32_t *__restrict__ M_dev, uint32_t M_host,
    const half8 *__restrict__ image, float loss_scale_host, half8 *__restrict__ d3buf, float *__restrict__ partial,
    float *__restrict__ ddirs, const int32_t *__restrict__ live_idx, const float *__restrict__ scaler, FieldAct act)
{
    extern __shared__ half8 lds_w[];   // fragments 0..45 (46 KiB); reused as the f32 reduction image at the end
    // scaler: the loss scale lives on the device and adapts (mlp_common.hpp: LossScalerWord); deltas then overflow to inf
    // instead of saturating -- the overflow is the signal
    const float loss_scale = scaler ? scaler[LS_SCALE] : loss_scale_host;
    const _Float16 lim = delta_limit(scaler);
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
    const uint32_t M = M_dev ? min((uint32_t)max(M_dev[0], 0), M_host) : M_host;
    const uint32_t n_tiles = (M + 31u) >> 5;
    for (uint32_t i = threadIdx.x; i < 46u * 64u; i += 256) lds_w[i] = image[i];
    __syncthreads();
    const half8 I0 = identity_frag(0, lane), I1 = identity_frag(1, lane);

    f32x16 g[8];   // 0,1: dW4[rb]   2..5: dW5[rb][cb]   6,7: dW6[cb]
#pragma unroll
    for (int i = 0; i < 8; i++) g[i] = zero16();

    NGP_STAMP_DECL;
    // live_idx: the kernel runs over a LIST of samples (M of them: the ones in front of the compositor's early stop, engine_
    // kernels.hip: live_index_kernel) instead of samples 0 .. M - 1; delta3 is stored in list order.  The list entry of a tile
    // is requested two tiles ahead, the tile's inputs one tile ahead.
    auto sample_of = [&](uint32_t t) -> uint32_t {
        const uint32_t c = t * 32u + n;
        return (t < n_tiles && c < M && live_idx) ? (uint32_t)live_idx[c] : c;
    };
    uint32_t row_cur = sample_of(wave), row_nxt = sample_of(wave + n_waves);
    RawTile nxt = fetch_raw<true>(enc, stride, dirs, dsigma, drgb, nullptr, row_cur, wave * 32u + n, wave * 32u + n < M, h);
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        asm volatile("" ::: "memory");   // keep the weight fragments in LDS: no hoisting of 46 KiB into VGPRs
        NGP_STAMP_BEGIN();
        const uint32_t c_idx = tile * 32u + n, row = row_cur;
        const bool valid = c_idx < M;
        const RawTile cur = nxt;
        {
            const uint32_t nc = (tile + n_waves) * 32u + n;
            nxt = fetch_raw<true>(enc, stride, dirs, dsigma, drgb, nullptr, row_nxt, nc, tile + n_waves < n_tiles && nc < M, h);
            row_cur = row_nxt;
            row_nxt = sample_of(tile + 2u * n_waves);
        }
        // samples behind the compositor's early stop (T < T_thresh) have exactly zero output gradients: a tile made of
        // such samples contributes nothing to any weight gradient and its encoder gradient is zero
        if (__ballot(cur.gs != 0.0f || cur.gr[0] != 0.0f || cur.gr[1] != 0.0f || cur.gr[2] != 0.0f) == 0ull) {
            if (valid) {
                half8 z;
#pragma unroll
                for (int t = 0; t < 8; t++) z[t] = (_Float16)0.0f;
                d3buf[(size_t)c_idx * 2 + h] = z;
                if (DDIRS && h == 0) ddirs[(size_t)row * 3] = ddirs[(size_t)row * 3 + 1] = ddirs[(size_t)row * 3 + 2] = 0.0f;
            }
            continue;
        }
        NGP_STAMP_AT(0);      // prefetch issue + skip test
        const TileInB<true> in = convert_raw<true>(cur, h);
        NGP_STAMP_AT(1);      // conversion of the tile's inputs (waits for last iteration's loads), SH

        // ---------------- recompute the forward pass
        f32x16 a[2];
        half8 x[2][2], h3[2][2], h4[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            a[rb] = zero16();
#pragma unroll
            for (int s = 0; s < 2; s++) a[rb] = mfma(NGP_FRAG(F_W1 + rb * 2 + s), in.x0[s], a[rb]);
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            x[kb][0] = pack_hidden<0, SOFT>(a[kb], act.beta);
            x[kb][1] = pack_hidden<1, SOFT>(a[kb], act.beta);
        }
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            a[rb] = zero16();
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int s = 0; s < 2; s++) a[rb] = mfma(NGP_FRAG(F_W2 + rb * 4 + kb * 2 + s), x[kb][s], a[rb]);
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            x[kb][0] = pack_hidden<0, SOFT>(a[kb], act.beta);
            x[kb][1] = pack_hidden<1, SOFT>(a[kb], act.beta);
        }
        f32x16 o = zero16();
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int s = 0; s < 2; s++) o = mfma(NGP_FRAG(F_W3 + kb * 2 + s), x[kb][s], o);
        const float sigma_raw = o[0];
        const half8 x3a = pack<0, false>(o);
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            a[rb] = zero16();
            a[rb] = mfma(NGP_FRAG(F_W4 + rb * 2 + 0), x3a, a[rb]);
            a[rb] = mfma(NGP_FRAG(F_W4 + rb * 2 + 1), in.sh, a[rb]);
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            h3[kb][0] = pack_hidden<0, SOFT>(a[kb], act.beta);
            h3[kb][1] = pack_hidden<1, SOFT>(a[kb], act.beta);
        }
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            a[rb] = zero16();
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int s = 0; s < 2; s++) a[rb] = mfma(NGP_FRAG(F_W5 + rb * 4 + kb * 2 + s), h3[kb][s], a[rb]);
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            h4[kb][0] = pack_hidden<0, SOFT>(a[kb], act.beta);
            h4[kb][1] = pack_hidden<1, SOFT>(a[kb], act.beta);
        }
        f32x16 c = zero16();
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int s = 0; s < 2; s++) c = mfma(NGP_FRAG(F_W6 + kb * 2 + s), h4[kb][s], c);

        NGP_STAMP_AT(2);      // forward recompute (32 MFMAs)

        // ---------------- output deltas (scaled by loss_scale so that they survive f16)
        const float gs = cur.gs, gr0 = cur.gr[0], gr1 = cur.gr[1], gr2 = cur.gr[2];
        f32x16 d6 = zero16();
        if (h == 0) {   // rows 0..2 of the tile: d rgb / d raw (default: exp(raw - 5) where the clamp at 5 is inactive)
            d6[0] = gr0 * act_dcolor(c[0], act) * loss_scale;
            d6[1] = gr1 * act_dcolor(c[1], act) * loss_scale;
            d6[2] = gr2 * act_dcolor(c[2], act) * loss_scale;
        }
        const half8 p6 = pack_sat<0>(d6, lim);

        // ---------------- layer 6: dW6 = delta6 x H4^T ; delta5 = W6^T delta6 (masked)
        half8 aT[2], bT[2];
        {
            const f32x16 t6 = mfma(p6, I0, zero16());
            aT[0] = pack<0, false>(t6);
            aT[1] = pack<1, false>(t6);
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                const f32x16 tt = transpose_tile(h4[cb][0], h4[cb][1], I0, I1);
                bT[0] = pack<0, false>(tt);
                bT[1] = pack<1, false>(tt);
                g[6 + cb] = mfma(aT[0], bT[0], g[6 + cb]);
                g[6 + cb] = mfma(aT[1], bT[1], g[6 + cb]);
            }
        }
        half8 p5[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            const f32x16 dh = mfma(NGP_FRAG(T_W6 + rb), p6, zero16());
            p5[rb][0] = pack_delta<0, SOFT>(dh, h4[rb][0], lim, act.beta);
            p5[rb][1] = pack_delta<1, SOFT>(dh, h4[rb][1], lim, act.beta);
        }
        NGP_STAMP_AT(3);      // deltas 6, dW6, delta 5
        // ---------------- layer 5: dW5 = delta5 x H3^T ; delta4 = W5^T delta5 (masked)
        {
            half8 a5[2][2];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const f32x16 tt = transpose_tile(p5[rb][0], p5[rb][1], I0, I1);
                a5[rb][0] = pack<0, false>(tt);
                a5[rb][1] = pack<1, false>(tt);
            }
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                const f32x16 tt = transpose_tile(h3[cb][0], h3[cb][1], I0, I1);
                bT[0] = pack<0, false>(tt);
                bT[1] = pack<1, false>(tt);
#pragma unroll
                for (int rb = 0; rb < 2; rb++) {
                    g[2 + rb * 2 + cb] = mfma(a5[rb][0], bT[0], g[2 + rb * 2 + cb]);
                    g[2 + rb * 2 + cb] = mfma(a5[rb][1], bT[1], g[2 + rb * 2 + cb]);
                }
            }
        }
        half8 p4[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            f32x16 dh = zero16();
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int s = 0; s < 2; s++) dh = mfma(NGP_FRAG(T_W5 + rb * 4 + kb * 2 + s), p5[kb][s], dh);
            p4[rb][0] = pack_delta<0, SOFT>(dh, h3[rb][0], lim, act.beta);
            p4[rb][1] = pack_delta<1, SOFT>(dh, h3[rb][1], lim, act.beta);
        }
        NGP_STAMP_AT(4);      // dW5, delta 4
        // ---------------- layer 4: dW4 = delta4 x X3^T ; dX3 = W4^T delta4
        {
            const f32x16 tx = transpose_tile(x3a, in.sh, I0, I1);
            bT[0] = pack<0, false>(tx);
            bT[1] = pack<1, false>(tx);
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const f32x16 tt = transpose_tile(p4[rb][0], p4[rb][1], I0, I1);
                aT[0] = pack<0, false>(tt);
                aT[1] = pack<1, false>(tt);
                g[rb] = mfma(aT[0], bT[0], g[rb]);
                g[rb] = mfma(aT[1], bT[1], g[rb]);
            }
        }
        f32x16 dx3 = zero16();
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int s = 0; s < 2; s++) dx3 = mfma(NGP_FRAG(T_W4 + kb * 2 + s), p4[kb][s], dx3);
        // delta3: rows 1..15 = d features, row 0 = d sigma_raw = dsigma * exp(clamp(raw, -80, 80))   (trunc_exp)
        if (h == 0) dx3[0] = gs * act_dsigma(sigma_raw, act) * loss_scale;
        if (valid) d3buf[(size_t)c_idx * 2 + h] = pack_sat<0>(dx3, lim);
        if constexpr (DDIRS) {
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
            // through u = d / |d| (applied twice by the reference -- renderer.py:541 and the encoder module: the same tangent
            // projection, idempotent); deltas carry the loss scale
            const float dot = gx * ux + gy * uy + gz * uz, k = inv / loss_scale;
            if (valid && h == 0) {
                ddirs[(size_t)row * 3] = (gx - ux * dot) * k;
                ddirs[(size_t)row * 3 + 1] = (gy - uy * dot) * k;
                ddirs[(size_t)row * 3 + 2] = (gz - uz * dot) * k;
            }
        }
        NGP_STAMP_AT(5);      // dW4, d x3, store
    }
    NGP_STAMP_BEGIN();
#ifdef NGP_STAMP_SPLIT_FLUSH   // diagnostic: where inside the flush
    __syncthreads();
    NGP_STAMP_AT(7);          // arrival skew at the first barrier
#endif
    flush_tiles_parallel<8>(reinterpret_cast<float *>(lds_w), g, lane, partial + (size_t)blockIdx.x * kAccFloats);
    NGP_STAMP_AT(6);          // partial-sum flush
    NGP_STAMP_FLUSH();
}

// ---- grid kernel ---------------------------------------------------------------------------------
// WINDOW: the encoder features are multiplied by a per-level weight before the MLP (BARF, network.py:99-109); the same
// weight then scales d enc
// UNIT: delta3 = e_0 for every sample instead of the view kernel's (d3buf NULL) and no weight gradients: d enc is then
// d h0 / d enc, the gradient of the density network's first output (sigma = trunc_exp(h0)) -- ngp_x_mlp_density_gradient
template <bool WINDOW, bool UNIT = false, bool SOFT = false>
__global__ __launch_bounds__(256, 1) void mlp_backward_grid_kernel(
    const float *__restrict__ enc, uint32_t stride, const int32_t *__restrict__ M_dev, uint32_t M_host,
    const half8 *__restrict__ image, float inv_loss_scale_host, const half8 *__restrict__ d3buf,
    float *__restrict__ denc, float *__restrict__ partial, const float *__restrict__ level_w, uint32_t t3_base,
    const int32_t *__restrict__ live_idx, const float *__restrict__ scaler, FieldAct act)
{
    extern __shared__ half8 lds_w[];   // local 0..11 = F_W1, F_W2 ; 12..25 = T_W3, T_W2, T_W1
    const float inv_loss_scale = scaler ? scaler[LS_INV] : inv_loss_scale_host;   // (see the view kernel)
    const _Float16 lim = delta_limit(scaler);
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
    const uint32_t M = M_dev ? min((uint32_t)max(M_dev[0], 0), M_host) : M_host;
    const uint32_t n_tiles = (M + 31u) >> 5;
    for (uint32_t i = threadIdx.x; i < 12u * 64u; i += 256) lds_w[i] = image[i];
    for (uint32_t i = threadIdx.x; i < 14u * 64u; i += 256) lds_w[12u * 64u + i] = image[(size_t)t3_base * 64 + i];
    __syncthreads();
    constexpr uint32_t LT_W3 = 12, LT_W2 = 14, LT_W1 = 22;
    const half8 I0 = identity_frag(0, lane), I1 = identity_frag(1, lane);
    LaneWindow lw;
    if constexpr (WINDOW) lw = load_window(level_w, h);

    f32x16 g[8];   // 0,1: dW1[rb]   2..5: dW2[rb][cb]   6,7: dW3[cb]
#pragma unroll
    for (int i = 0; i < 8; i++) g[i] = zero16();

    // live_idx: as in the view kernel -- the encoder rows of sample live_idx[c], delta3 and d enc in list order (c)
    auto sample_of = [&](uint32_t t) -> uint32_t {
        const uint32_t c = t * 32u + n;
        return (t < n_tiles && c < M && live_idx) ? (uint32_t)live_idx[c] : c;
    };
    uint32_t row_nxt = sample_of(wave + n_waves);
    RawTile nxt = fetch_raw<false>(enc, stride, nullptr, nullptr, nullptr, d3buf, sample_of(wave), wave * 32u + n,
                                   wave * 32u + n < M, h);
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        asm volatile("" ::: "memory");   // keep the weight fragments in LDS (see the view kernel)
        const uint32_t row = tile * 32u + n;   // (list order: where delta3 is read and d enc written)
        const bool valid = row < M;
        const RawTile cur = nxt;
        {
            const uint32_t nc = (tile + n_waves) * 32u + n;
            nxt = fetch_raw<false>(enc, stride, nullptr, nullptr, nullptr, d3buf, row_nxt, nc, tile + n_waves < n_tiles && nc < M, h);
            row_nxt = sample_of(tile + 2u * n_waves);
        }
        half8 p3 = cur.p3;
        if constexpr (UNIT) {
            if (valid && h == 0) p3[0] = (_Float16)1.0f;
        }
        {   // all-zero deltas (see the view kernel): the tile's encoder gradient is zero, nothing else changes
            typedef short short8 __attribute__((ext_vector_type(8)));
            const short8 bits = __builtin_bit_cast(short8, p3) & (short8)0x7fff;   // -0 counts as zero
            const bool nz = (bits[0] | bits[1] | bits[2] | bits[3] | bits[4] | bits[5] | bits[6] | bits[7]) != 0;
            if (__ballot(nz) == 0ull) {
                if (valid) {
#pragma unroll
                    for (uint32_t q = 0; q < 4; q++) {
                        const uint32_t level = 4 * q + 2 * h;
                        reinterpret_cast<float2 *>(denc)[(size_t)level * stride + row] = make_float2(0.f, 0.f);
                        reinterpret_cast<float2 *>(denc)[(size_t)(level + 1) * stride + row] = make_float2(0.f, 0.f);
                    }
                }
                continue;
            }
        }
        TileInB<false> in;
        if constexpr (WINDOW) {
            RawTile scaled = cur;
#pragma unroll
            for (int i = 0; i < 8; i++) {   // r.e[4s + 2q + e] = level 8s + 4q + 2h + e: the window's own order
                scaled.e[i].x *= lw.w[i];
                scaled.e[i].y *= lw.w[i];
            }
            in = convert_raw<false>(scaled, h);
        } else {
            in = convert_raw<false>(cur, h);
        }

        f32x16 a[2];
        half8 h1[2][2], h2[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            a[rb] = zero16();
#pragma unroll
            for (int s = 0; s < 2; s++) a[rb] = mfma(NGP_FRAG(F_W1 + rb * 2 + s), in.x0[s], a[rb]);
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            h1[kb][0] = pack_hidden<0, SOFT>(a[kb], act.beta);
            h1[kb][1] = pack_hidden<1, SOFT>(a[kb], act.beta);
        }
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            a[rb] = zero16();
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int s = 0; s < 2; s++) a[rb] = mfma(NGP_FRAG(F_W2 + rb * 4 + kb * 2 + s), h1[kb][s], a[rb]);
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            h2[kb][0] = pack_hidden<0, SOFT>(a[kb], act.beta);
            h2[kb][1] = pack_hidden<1, SOFT>(a[kb], act.beta);
        }

        half8 aT[2], bT[2];
        // ---------------- layer 3: dW3 = delta3 x H2^T ; delta2 = W3^T delta3 (masked)
        if constexpr (!UNIT) {
            const f32x16 t3 = mfma(p3, I0, zero16());
            aT[0] = pack<0, false>(t3);
            aT[1] = pack<1, false>(t3);
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                const f32x16 tt = transpose_tile(h2[cb][0], h2[cb][1], I0, I1);
                bT[0] = pack<0, false>(tt);
                bT[1] = pack<1, false>(tt);
                g[6 + cb] = mfma(aT[0], bT[0], g[6 + cb]);
                g[6 + cb] = mfma(aT[1], bT[1], g[6 + cb]);
            }
        }
        half8 p2[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            const f32x16 dh = mfma(NGP_FRAG(LT_W3 + rb), p3, zero16());
            p2[rb][0] = pack_delta<0, SOFT>(dh, h2[rb][0], lim, act.beta);
            p2[rb][1] = pack_delta<1, SOFT>(dh, h2[rb][1], lim, act.beta);
        }
        // ---------------- layer 2: dW2 = delta2 x H1^T ; delta1 = W2^T delta2 (masked)
        if constexpr (!UNIT) {
            half8 a2[2][2];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const f32x16 tt = transpose_tile(p2[rb][0], p2[rb][1], I0, I1);
                a2[rb][0] = pack<0, false>(tt);
                a2[rb][1] = pack<1, false>(tt);
            }
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                const f32x16 tt = transpose_tile(h1[cb][0], h1[cb][1], I0, I1);
                bT[0] = pack<0, false>(tt);
                bT[1] = pack<1, false>(tt);
#pragma unroll
                for (int rb = 0; rb < 2; rb++) {
                    g[2 + rb * 2 + cb] = mfma(a2[rb][0], bT[0], g[2 + rb * 2 + cb]);
                    g[2 + rb * 2 + cb] = mfma(a2[rb][1], bT[1], g[2 + rb * 2 + cb]);
                }
            }
        }
        half8 p1[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            f32x16 dh = zero16();
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int s = 0; s < 2; s++) dh = mfma(NGP_FRAG(LT_W2 + rb * 4 + kb * 2 + s), p2[kb][s], dh);
            p1[rb][0] = pack_delta<0, SOFT>(dh, h1[rb][0], lim, act.beta);
            p1[rb][1] = pack_delta<1, SOFT>(dh, h1[rb][1], lim, act.beta);
        }
        // ---------------- layer 1: dW1 = delta1 x X0^T ; d enc = W1^T delta1
        if constexpr (!UNIT) {
            const f32x16 tx = transpose_tile(in.x0[0], in.x0[1], I0, I1);
            bT[0] = pack<0, false>(tx);
            bT[1] = pack<1, false>(tx);
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const f32x16 tt = transpose_tile(p1[rb][0], p1[rb][1], I0, I1);
                aT[0] = pack<0, false>(tt);
                aT[1] = pack<1, false>(tt);
                g[rb] = mfma(aT[0], bT[0], g[rb]);
                g[rb] = mfma(aT[1], bT[1], g[rb]);
            }
        }
        f32x16 dx0 = zero16();
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int s = 0; s < 2; s++) dx0 = mfma(NGP_FRAG(LT_W1 + kb * 2 + s), p1[kb][s], dx0);
        if (valid) {
            // registers 4q..4q+3 = features 8q + 4h + {0..3} = levels 4q + 2h, 4q + 2h + 1 (both channels)
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                const uint32_t level = 4 * q + 2 * h;
                float klo = inv_loss_scale, khi = inv_loss_scale;
                if constexpr (WINDOW) {   // levels 4q + 2h + {0, 1} = window entries (s = q >> 1, q & 1, e)
                    klo *= lw.w[4 * (q >> 1) + 2 * (q & 1)];
                    khi *= lw.w[4 * (q >> 1) + 2 * (q & 1) + 1];
                }
                float2 lo = make_float2(dx0[4 * q] * klo, dx0[4 * q + 1] * klo);
                float2 hi = make_float2(dx0[4 * q + 2] * khi, dx0[4 * q + 3] * khi);
                reinterpret_cast<float2 *>(denc)[(size_t)level * stride + row] = lo;
                reinterpret_cast<float2 *>(denc)[(size_t)(level + 1) * stride + row] = hi;
            }
        }
    }
    if constexpr (!UNIT)
        flush_tiles_parallel<8>(reinterpret_cast<float *>(lds_w), g, lane, partial + (size_t)blockIdx.x * kAccFloats);
}

// ---- partial-slab reduction -----------------------------------------------------------------------
// element e of a workgroup slab: tile b = e / 1024, register v = (e / 64) % 16, lane = e % 64
// -> dW[32*rb + o][32*cb + j] with o = (v&3) + 8(v>>2) + 4(lane>>5), j = lane & 31
// optional: torch.optim.Adam (as engine_kernels.hip: adam_span) on the flat MLP weight buffer, element by element as the
// gradients come out of the reduction; dw1..dw6 must then be views of `grad`
__global__ __launch_bounds__(256) void mlp_reduce_dw_kernel(MlpDwReduce r)
{
    __shared__ float part[4][64];
    mlp_reduce_dw_group(r, blockIdx.x, threadIdx.x, part);
}
// dynamic loss scale: Adam on the MLP weights as a launch of its own, after the reduction has seen every weight gradient
__global__ __launch_bounds__(256) void mlp_adam_kernel(MlpDwReduce r)
{
    if (reinterpret_cast<const uint32_t *>(r.scaler)[LS_FOUND] != 0u) return;   // the step is skipped
    mlp_adam_group(r, blockIdx.x * 256u + threadIdx.x);
}

// the backward kernels stage their partial sums in 64 KiB of dynamic LDS (the default limit)
static bool mlp_backward_lds_ok()
{
    static const bool ok = [] {
        bool r = hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_view_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kViewLds) == hipSuccess;
        r = r && hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_view_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kViewLds) == hipSuccess;
        r = r && hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_grid_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGridLds) == hipSuccess;
        r = r && hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_grid_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGridLds) == hipSuccess;
        r = r && hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_grid_kernel<false, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGridLds) == hipSuccess;
        r = r && hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_grid_kernel<true, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGridLds) == hipSuccess;
        r = r && hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_grid_kernel<false, false, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGridLds) == hipSuccess;
        r = r && hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_view_kernel<false, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kViewLds) == hipSuccess;
        r = r && hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_view_kernel<true, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kViewLds) == hipSuccess;
        return r;
    }();
    return ok;
}

int launch_mlp_backward_grid(const float *enc, uint32_t stride, const float *level_w, const int32_t *M_dev, uint32_t M,
                             const half8 *image, uint32_t t3_base, float inv_loss_scale, const half8 *d3buf, float *denc,
                             float *partial, uint32_t blocks, hipStream_t st, const int32_t *sample_index,
                             const float *scaler, FieldAct act)
{
    NGP_REQUIRE(mlp_backward_lds_ok(), "mlp_backward_grid: cannot raise the dynamic LDS limit");
    NGP_REQUIRE(!(level_w && act.internal), "mlp_backward_grid: the level window comes with ReLU hidden layers only");
    if (act.internal)
        mlp_backward_grid_kernel<false, false, true><<<dim3(blocks), dim3(256), kGridLds, st>>>(
            enc, stride, M_dev, M, image, inv_loss_scale, d3buf, denc, partial, nullptr, t3_base, sample_index, scaler, act);
    else if (level_w)
        mlp_backward_grid_kernel<true><<<dim3(blocks), dim3(256), kGridLds, st>>>(enc, stride, M_dev, M, image, inv_loss_scale,
                                                                                  d3buf, denc, partial, level_w, t3_base,
                                                                                  sample_index, scaler, act);
    else
        mlp_backward_grid_kernel<false><<<dim3(blocks), dim3(256), kGridLds, st>>>(enc, stride, M_dev, M, image, inv_loss_scale,
                                                                                   d3buf, denc, partial, nullptr, t3_base,
                                                                                   sample_index, scaler, act);
    NGP_CHECK_LAUNCH("mlp_backward_grid");
    return NGP_OK;
}

}  // namespace ngp

using namespace ngp;

#ifdef NGP_STAMP
extern "C" int ngp_dbg_read_stamps(unsigned long long *out16, int reset)
{
    unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ngp_dbg_stamps), sizeof(zero)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ngp_dbg_stamps), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

static uint32_t mlp_bwd_blocks(uint32_t M)
{
    const uint32_t tiles = ceil_div(M, 32u);
    return max(1u, min(ceil_div(tiles, 4u), 256u));
}

// arguments of the weight-gradient reduction over the partial slabs ngp_x_mlp_backward(M, ...) left in `workspace`
// (checks as ngp_x_mlp_reduce_dw; also used by the table backward's launch, which can carry this reduction along)
int ngp::mlp_dw_reduce_args(MlpDwReduce &r, const char *who, uint32_t M, float loss_scale, float *dw1, float *dw2, float *dw3,
                            float *dw4, float *dw5, float *dw6, const void *workspace, size_t workspace_bytes,
                            float *adam_param, const float *adam_grad, float *adam_exp_avg, float *adam_exp_avg_sq,
                            uint32_t adam_n, const float *adam_hyper, float beta1, float beta2, float eps, void *adam_image,
                            float *scaler)
{
    NGP_REQUIRE(dw1 && dw2 && dw3 && dw4 && dw5 && dw6 && workspace, "%s: null tensor", who);
    NGP_REQUIRE(!adam_image || adam_param, "%s: the operand image is only patched together with Adam", who);
    if (adam_param) {
        NGP_REQUIRE(adam_grad && adam_exp_avg && adam_exp_avg_sq && adam_hyper, "%s: incomplete Adam state", who);
        for (const float *d : {dw1, dw2, dw3, dw4, dw5, dw6})
            NGP_REQUIRE(d >= adam_grad && d < adam_grad + adam_n, "%s: dw tensors must be views of adam_grad", who);
    }
    NGP_REQUIRE(workspace_bytes >= ngp_x_mlp_backward_workspace_bytes(M), "%s: workspace too small", who);
    NGP_REQUIRE(loss_scale > 0.0f, "%s: loss_scale must be positive", who);
    r.part_view = reinterpret_cast<const float *>(reinterpret_cast<const char *>(workspace) + (((size_t)M * 32 + 255) & ~(size_t)255));
    r.part_grid = r.part_view + (size_t)256 * kAccFloats;
    r.n_wg = mlp_bwd_blocks(max(M, 1u));
    r.inv_loss_scale = 1.0f / loss_scale;
    r.dw1 = dw1, r.dw2 = dw2, r.dw3 = dw3, r.dw4 = dw4, r.dw5 = dw5, r.dw6 = dw6;
    r.adam = MlpAdam{adam_param, adam_grad, adam_exp_avg, adam_exp_avg_sq, adam_hyper, beta1, beta2, eps,
                     reinterpret_cast<_Float16 *>(adam_image)};
    r.scaler = scaler;
    return NGP_OK;
}

extern "C" size_t ngp_x_mlp_backward_workspace_bytes(uint32_t M)
{
    return (size_t)M * 32 + 2 * (size_t)256 * kAccFloats * 4 + 256;
}

extern "C" int ngp_x_mlp_backward(const float *enc, uint32_t stride, const float *dirs, const float *dsigma,
                                  const float *drgb, const int32_t *M_dev, uint32_t M, const void *image,
                                  float loss_scale, float *denc, float *dw1, float *dw2, float *dw3, float *dw4,
                                  float *dw5, float *dw6, void *workspace, size_t workspace_bytes, ngp_stream_t stream)
{
    return ngp_x_mlp_backward_dirs(enc, stride, dirs, dsigma, drgb, M_dev, M, image, loss_scale, denc, nullptr, dw1, dw2, dw3,
                                   dw4, dw5, dw6, workspace, workspace_bytes, stream);
}

extern "C" int ngp_x_mlp_backward_dirs(const float *enc, uint32_t stride, const float *dirs, const float *dsigma,
                                       const float *drgb, const int32_t *M_dev, uint32_t M, const void *image,
                                       float loss_scale, float *denc, float *ddirs, float *dw1, float *dw2, float *dw3,
                                       float *dw4, float *dw5, float *dw6, void *workspace, size_t workspace_bytes,
                                       ngp_stream_t stream)
{
    return ngp_x_mlp_backward_list(enc, stride, dirs, dsigma, drgb, M_dev, M, nullptr, image, loss_scale, denc, ddirs, dw1, dw2,
                                   dw3, dw4, dw5, dw6, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int ngp_x_mlp_backward_list(const float *enc, uint32_t stride, const float *dirs, const float *dsigma,
                                       const float *drgb, const int32_t *M_dev, uint32_t M, const int32_t *sample_index,
                                       const void *image, float loss_scale, float *denc, float *ddirs, float *dw1,
                                       float *dw2, float *dw3, float *dw4, float *dw5, float *dw6, void *workspace,
                                       size_t workspace_bytes, float *loss_scaler, ngp_stream_t stream)
{
    return ngp_x_mlp_backward_act(enc, stride, dirs, dsigma, drgb, M_dev, M, sample_index, image, loss_scale, denc, ddirs, dw1,
                                  dw2, dw3, dw4, dw5, dw6, workspace, workspace_bytes, loss_scaler, 0, 0, 0, 1.0f, stream);
}

// ... with the field's non-default output activations (as ngp_x_mlp_forward_act)
extern "C" int ngp_x_mlp_backward_act(const float *enc, uint32_t stride, const float *dirs, const float *dsigma,
                                      const float *drgb, const int32_t *M_dev, uint32_t M, const int32_t *sample_index,
                                      const void *image, float loss_scale, float *denc, float *ddirs, float *dw1,
                                      float *dw2, float *dw3, float *dw4, float *dw5, float *dw6, void *workspace,
                                      size_t workspace_bytes, float *loss_scaler, uint32_t color_act, uint32_t density_act,
                                      uint32_t internal_act, float beta, ngp_stream_t stream)
{
    NGP_REQUIRE(color_act <= 2u && density_act <= 1u && internal_act <= 1u && beta > 0.0f,
                "mlp_backward: unknown activation or beta <= 0");
    FieldAct act;
    act.color = color_act, act.density = density_act, act.beta = beta, act.internal = internal_act;
    const bool reduce_now = dw1 != nullptr;   // all NULL: leave the per-workgroup partials for ngp_x_mlp_reduce_dw
    NGP_REQUIRE(image && workspace, "mlp_backward: null tensor");
    NGP_REQUIRE(reduce_now ? (dw2 && dw3 && dw4 && dw5 && dw6) : (!dw2 && !dw3 && !dw4 && !dw5 && !dw6),
                "mlp_backward: pass all six weight-gradient tensors or none");
    NGP_REQUIRE(M == 0 || (enc && dirs && dsigma && drgb && denc), "mlp_backward: null sample tensor");
    NGP_REQUIRE(stride >= M, "mlp_backward: encoder slab stride smaller than M");
    NGP_REQUIRE(workspace_bytes >= ngp_x_mlp_backward_workspace_bytes(M), "mlp_backward: workspace too small");
    NGP_REQUIRE(((uintptr_t)workspace & 15u) == 0, "mlp_backward: workspace must be 16-byte aligned");
    NGP_REQUIRE(loss_scale > 0.0f, "mlp_backward: loss_scale must be positive");
    hipStream_t st = as_stream(stream);
    NGP_REQUIRE(mlp_backward_lds_ok(), "mlp_backward: cannot raise the dynamic LDS limit");
    const uint32_t blocks = mlp_bwd_blocks(max(M, 1u));
    half8 *d3buf = reinterpret_cast<half8 *>(workspace);
    float *part_view = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + (((size_t)M * 32 + 255) & ~(size_t)255));
    float *part_grid = part_view + (size_t)256 * kAccFloats;
    const half8 *img = reinterpret_cast<const half8 *>(image);
#define NGP_VIEW_ARGS enc, stride, dirs, dsigma, drgb, M_dev, M, img, loss_scale, d3buf, part_view, ddirs, sample_index, loss_scaler, act
    if (ddirs && act.internal)
        mlp_backward_view_kernel<true, true><<<dim3(blocks), dim3(256), kViewLds, st>>>(NGP_VIEW_ARGS);
    else if (ddirs)
        mlp_backward_view_kernel<true><<<dim3(blocks), dim3(256), kViewLds, st>>>(NGP_VIEW_ARGS);
    else if (act.internal)
        mlp_backward_view_kernel<false, true><<<dim3(blocks), dim3(256), kViewLds, st>>>(NGP_VIEW_ARGS);
    else
        mlp_backward_view_kernel<false><<<dim3(blocks), dim3(256), kViewLds, st>>>(NGP_VIEW_ARGS);
#undef NGP_VIEW_ARGS
    {
        const int rc = launch_mlp_backward_grid(enc, stride, nullptr, M_dev, M, img, T_W3, 1.0f / loss_scale, d3buf, denc, part_grid,
                                                blocks, st, sample_index, loss_scaler, act);
        if (rc != NGP_OK) return rc;
    }
    if (reduce_now)
        mlp_reduce_dw_kernel<<<dim3(kDwGroups), dim3(256), 0, st>>>(
            MlpDwReduce{part_view, part_grid, blocks, 1.0f / loss_scale, dw1, dw2, dw3, dw4, dw5, dw6, MlpAdam{}, loss_scaler});
    NGP_CHECK_LAUNCH("mlp_backward");
    return NGP_OK;
}

// d h0 / d enc for every sample, h0 = the density network's first output (sigma = trunc_exp(h0), network.py:111-118): what
// torch.autograd.grad(sigma, pos) of the orientation term (renderer.py:558-566) needs from the MLP, up to the factor
// d sigma / d h0 the caller knows from sigma itself.  Same f16 chain as the backward's density kernel, delta3 = e_0.
namespace ngp {
int launch_mlp_density_gradient(const char *who, const float *enc, uint32_t stride, const float *level_w, const int32_t *M_dev,
                                uint32_t M, const void *image, uint32_t t3_base, float *denc, hipStream_t st)
{
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(enc && image && denc, "%s: null tensor", who);
    NGP_REQUIRE(stride >= M, "%s: encoder slab stride smaller than M", who);
    NGP_REQUIRE(mlp_backward_lds_ok(), "%s: cannot raise the dynamic LDS limit", who);
    if (level_w)    // the window scales the features in front of the network and, as its adjoint, d enc behind it
        mlp_backward_grid_kernel<true, true><<<dim3(mlp_bwd_blocks(M)), dim3(256), kGridLds, st>>>(
            enc, stride, M_dev, M, reinterpret_cast<const half8 *>(image), 1.0f, nullptr, denc, nullptr, level_w, t3_base,
            nullptr, nullptr, FieldAct{});
    else
        mlp_backward_grid_kernel<false, true><<<dim3(mlp_bwd_blocks(M)), dim3(256), kGridLds, st>>>(
            enc, stride, M_dev, M, reinterpret_cast<const half8 *>(image), 1.0f, nullptr, denc, nullptr, nullptr, t3_base,
            nullptr, nullptr, FieldAct{});
    NGP_CHECK_LAUNCH(who);
    return NGP_OK;
}
}  // namespace ngp

extern "C" int ngp_x_mlp_density_gradient(const float *enc, uint32_t stride, const int32_t *M_dev, uint32_t M,
                                          const void *image, float *denc, ngp_stream_t stream)
{
    return launch_mlp_density_gradient("mlp_density_gradient", enc, stride, nullptr, M_dev, M, image, T_W3, denc,
                                       as_stream(stream));
}

// second half of ngp_x_mlp_backward when it was called without weight-gradient tensors: sum the per-workgroup
// partial slabs the two backward kernels left in `workspace` (same M, same loss_scale)
extern "C" int ngp_x_mlp_reduce_dw(uint32_t M, float loss_scale, float *dw1, float *dw2, float *dw3, float *dw4,
                                   float *dw5, float *dw6, const void *workspace, size_t workspace_bytes,
                                   float *adam_param, const float *adam_grad, float *adam_exp_avg, float *adam_exp_avg_sq,
                                   uint32_t adam_n, const float *adam_hyper, float beta1, float beta2, float eps,
                                   void *adam_image, float *loss_scaler, ngp_stream_t stream)
{
    MlpDwReduce r;
    const int rc = mlp_dw_reduce_args(r, "mlp_reduce_dw", M, loss_scale, dw1, dw2, dw3, dw4, dw5, dw6, workspace, workspace_bytes,
                                      adam_param, adam_grad, adam_exp_avg, adam_exp_avg_sq, adam_n, adam_hyper, beta1, beta2,
                                      eps, adam_image, loss_scaler);
    if (rc != NGP_OK) return rc;
    mlp_reduce_dw_kernel<<<dim3(kDwGroups), dim3(256), 0, as_stream(stream)>>>(r);
    if (loss_scaler && adam_param)   // the optimiser step waits for the verdict over ALL weight gradients
        mlp_adam_kernel<<<dim3(2 * kAccFloats / 256), dim3(256), 0, as_stream(stream)>>>(r);
    NGP_CHECK_LAUNCH("mlp_reduce_dw");
    return NGP_OK;
}
