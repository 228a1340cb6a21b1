// Kernels of the fused step for the pose-refinement / light-conditioned configuration (BASELINE
// configs[3]).  What the reference does around its kernels with torch ops -- barf/camera.py (se(3) exponential, pose
// composition), barf/camera_optimizers.py:14-52,94-106 (one correction per camera, Adam with an exponential decay),
// nerf/network.py:99-109 (BARF level window from the annealing value), raymarching/raymarching.py:319-329 (ray gradients
// as segment sums of the sample gradients), gridencoder/src/gridencoder.cu:352-378 (input gradient) -- as a handful of
// launches with no host round trip, so that the whole step can be replayed from a hipGraph:
//
//   step_window     annealing = float16(step / iters) (train_utils.py:488) -> the 16 level weights, and whether this is
//                   still a pose step (annealing < end_annealing, train_utils.py:901)
//   ray_gradients   per ray: sum over its samples of d xyz (= sum_l d enc_l . dy_dx_l / (2 bound), the encoder's input
//                   backward, never materialised) and of t * d xyz + d dirs
//   pose_gradient   per camera: d loss / d (refined pose) from the ray gradients of the rays drawn from that camera
//   pose_update     per camera: chain through compose(exp(xi), base) by forward-mode differentiation of the exponential
//                   map, torch.optim.Adam on xi, the refined pose of the next step
#include "grid_common.hpp"

namespace ngp {

// ------------------------------------------------------------------ annealing -> level window
// The reference keeps the annealing value as numpy float16 and evaluates alpha = (annealing - start) / (end - start) * L in
// that type (every operation rounds to half precision; the python floats are cast to float16 first), then the cosine
// window in float32 torch ops (network.py:101-108).  level_w[0] is forced to 1 (weights[0:2] = 1).
// flags[0] = 1 while annealing < end_annealing (the pose optimiser steps), flags[1] = steps done before this one (the
// pose optimiser's own step index).  The reference increments global_step BEFORE train_step (train_utils.py:887-888) and
// forms annealing from the incremented value (:488): the training step with s steps behind it sees (s + 1) / iters --
// step_offset = 1 -- while the density-grid refresh in front of it still sees the previous step's window (s / iters; 0.0
// before the first step, :411).
// BAA: the BAA-NGP window (network.py:77-97): level 0 always counts, level j >= 1 ramps in like the (j - 1)-th of L - 1 levels
// (alpha is scaled by L - 1 = grid_mlp.dim_out - 1, :78-83); what the window does to the features is slab_window_kernel's job
template <bool BAA>
__global__ void step_window_kernel(const uint32_t *__restrict__ step_counter, uint32_t step_offset, double iters,
                                   float start, float end, uint32_t L, float *__restrict__ level_w,
                                   int32_t *__restrict__ flags)
{
    const uint32_t k = threadIdx.x;
    const uint32_t step = step_counter[0] + step_offset;
    const _Float16 ann = (_Float16)fmin(fmax((double)step / iters, 0.0), 1.0);
    // `if end == 0: end = 1e-12` (network.py:103-104); end - start is python float arithmetic, cast to float16 as a whole
    const double span = (end == 0.0f ? 1e-12 : (double)end) - (double)start;
    const _Float16 e16 = (_Float16)(double)end;                     // the pose-step test uses opt.end_annealing as given
    const uint32_t Lw = BAA ? L - 1u : L;
    float alpha = (float)(((ann - (_Float16)(double)start) / (_Float16)span) * (_Float16)(float)Lw);
    // end == 0 ("no annealing"): the 1e-12 guard underflows to 0 in float16; what the guard is there for -- and what
    // float64 arithmetic (NumPy 1.x scalar promotion) gives -- is a window that is fully open from the first step on
    if (end == 0.0f) alpha = (float)(((double)(float)ann - (double)start) / span * (double)Lw);
    if (k < L) {
        const float x = fminf(fmaxf(alpha - (float)(BAA ? k - 1u : k), 0.0f), 1.0f);
        level_w[k] = k == 0 ? 1.0f : (1.0f - cosf(x * 3.14159265358979323846f)) / 2.0f;
    }
    if (k == 0 && flags) {
        flags[0] = ann < e16 ? 1 : 0;
        flags[1] = (int32_t)step_counter[0];
    }
}

// ------------------------------------------------------------------ BAA-NGP window on the encoder slab
// network.py:84-97: with c = the finest level whose weight is > 0 (its two features are `coarse_features`),
//   f'_l = w_l f_l + (1 - w_l) f_c          for every level l (f'_c = f_c)
// applied in place to the level-major slab enc[L][stride][2] before the field reads it.  BWD: the adjoint, in place on
// d enc:  d f_l = w_l d f'_l  (l != c),   d f_c = d f'_c + sum_{l != c} (1 - w_l) d f'_l.   One lane per sample.
// MODE 0: blend, 1: its adjoint, 2: plain per-level scale (the BARF window; self-adjoint)
template <int MODE>
__global__ __launch_bounds__(256) void slab_window_kernel(float2 *__restrict__ slab, uint32_t stride, uint32_t L,
                                                          const float *__restrict__ level_w,
                                                          const int32_t *__restrict__ M_dev, uint32_t M)
{
    constexpr bool BWD = MODE == 1;
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = M_dev ? min((uint32_t)max(M_dev[0], 0), M) : M;
    if (b >= n) return;
    if (MODE == 2) {
        for (uint32_t l = 0; l < L; l++) {
            const float w = level_w[l];
            float2 f = slab[(size_t)l * stride + b];
            f.x *= w;
            f.y *= w;
            slab[(size_t)l * stride + b] = f;
        }
        return;
    }
    uint32_t c = 0;
    for (uint32_t l = 0; l < L; l++)
        if (level_w[l] > 0.0f) c = l;
    if (!BWD) {
        const float2 fc = slab[(size_t)c * stride + b];
        for (uint32_t l = 0; l < L; l++) {
            if (l == c) continue;
            const float w = level_w[l];
            float2 f = slab[(size_t)l * stride + b];
            f.x = f.x * w + fc.x * (1.0f - w);
            f.y = f.y * w + fc.y * (1.0f - w);
            slab[(size_t)l * stride + b] = f;
        }
    } else {
        float2 gc = slab[(size_t)c * stride + b];
        for (uint32_t l = 0; l < L; l++) {
            if (l == c) continue;
            const float w = level_w[l];
            float2 g = slab[(size_t)l * stride + b];
            gc.x += (1.0f - w) * g.x;
            gc.y += (1.0f - w) * g.y;
            g.x *= w;
            g.y *= w;
            slab[(size_t)l * stride + b] = g;
        }
        slab[(size_t)c * stride + b] = gc;
    }
}

// ------------------------------------------------------------------ ray gradients
// one wave per ray; lanes stride over the ray's samples.  denc / dydx are level-major slabs ([L][stride][2], [L][stride][3][2]);
// ddirs may be NULL.  Sums are formed in a fixed order (lane partials, then a butterfly): reproducible.
__global__ __launch_bounds__(256) void ray_gradients_kernel(const float *__restrict__ denc, const float *__restrict__ dydx,
                                                            uint32_t stride, uint32_t L, float inv_2bound,
                                                            const float *__restrict__ ddirs, const float *__restrict__ ts,
                                                            const int32_t *__restrict__ rays, uint32_t N, uint32_t M,
                                                            float *__restrict__ grad_rays_o, float *__restrict__ grad_rays_d,
                                                            const int32_t *__restrict__ live_n,
                                                            const int32_t *__restrict__ live_off,
                                                            const float *__restrict__ term_weight = nullptr,
                                                            const float *__restrict__ dterm_ddirs = nullptr)
{
    const uint32_t n = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[(size_t)n * 2], cnt = (uint32_t)rays[(size_t)n * 2 + 1];
    // live_n / live_off: the backward ran over the LIST of samples in front of the compositor's early stop (engine_kernels.hip:
    // live_index_kernel) -- d enc is in list order (this ray's entries start at live_off[n]), only the ray's first live_n[n]
    // samples have gradients at all (d dirs of the others was not written)
    const uint32_t take = live_n ? min((uint32_t)live_n[n], cnt) : cnt;
    const size_t at = live_off ? (size_t)(uint32_t)live_off[n] : (size_t)off;
    float so[3] = {0, 0, 0}, sd[3] = {0, 0, 0};
    if (off + cnt <= M) {
        for (uint32_t k = lane; k < take; k += kWave) {
            const size_t i = (size_t)off + k, ic = at + k;
            float g[3] = {0, 0, 0};
            // eight levels' loads leave before the first multiply-add waits (32 requests in flight per lane -- all sixteen levels at once measure slower, 46 against 43 us; taken level by level
            // the wave paid a memory round trip per level: 70 us for 262 k samples); the additions keep their order
            uint32_t l0 = 0;
            for (; l0 + 8u <= L; l0 += 8u) {
                float2 ge[8], jj[8][3];
#pragma unroll
                for (uint32_t u = 0; u < 8u; u++) {
                    ge[u] = reinterpret_cast<const float2 *>(denc)[(size_t)(l0 + u) * stride + ic];
                    const float2 *j = reinterpret_cast<const float2 *>(dydx) + ((size_t)(l0 + u) * stride + i) * 3;
#pragma unroll
                    for (int d = 0; d < 3; d++) jj[u][d] = j[d];
                }
#pragma unroll
                for (uint32_t u = 0; u < 8u; u++) {
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        g[d] = fmaf(ge[u].x, jj[u][d].x, g[d]);
                        g[d] = fmaf(ge[u].y, jj[u][d].y, g[d]);
                    }
                }
            }
            for (uint32_t l = l0; l < L; l++) {
                const float2 ge = reinterpret_cast<const float2 *>(denc)[(size_t)l * stride + ic];
                const float2 *j = reinterpret_cast<const float2 *>(dydx) + ((size_t)l * stride + i) * 3;
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const float2 jj = j[d];
                    g[d] = fmaf(ge.x, jj.x, g[d]);
                    g[d] = fmaf(ge.y, jj.y, g[d]);
                }
            }
            const float t = ts[i * 2];      // the reference multiplies by ts[:, 0] (raymarching.py:297,328)
            // a loss term over the samples' weights that depends on the view direction (the orientation term):
            // term_weight[i] = lambda * weights[i] from the compositor step, dterm_ddirs from ngp_x_orientation_term
            const float tw = term_weight ? term_weight[i] : 0.0f;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const float gx = g[d] * inv_2bound;
                so[d] += gx;
                sd[d] += gx * t + (ddirs ? ddirs[i * 3 + d] : 0.0f) + (term_weight ? tw * dterm_ddirs[i * 3 + d] : 0.0f);
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

// ------------------------------------------------------------------ orientation term, per sample
// renderer.py:558-571: normals = -normalize(d sigma / d xyz) mapped to [0, 1], n_dot_v against the (normalised) view
// direction, min(0, .)^2.  d sigma / d xyz = trunc_exp'(h0) * sum_l (d h0 / d enc_l) . (d enc_l / d x01) / (2 bound); the
// first factor only matters against normalize's eps.  One thread per sample.
__global__ __launch_bounds__(256) void orientation_term_kernel(const float *__restrict__ dh_denc, const float *__restrict__ dydx,
                                                               uint32_t stride, uint32_t L, float inv_2bound,
                                                               const float *__restrict__ sigmas, const float *__restrict__ dirs,
                                                               const int32_t *__restrict__ M_dev, uint32_t M_cap,
                                                               float *__restrict__ term, float *__restrict__ dterm_ddirs,
                                                               float softplus_beta)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t M = M_dev ? min((uint32_t)max(M_dev[0], 0), M_cap) : M_cap;
    if (i >= M) return;
    float g[3] = {0, 0, 0};
    for (uint32_t l = 0; l < L; l++) {
        const float2 ge = reinterpret_cast<const float2 *>(dh_denc)[(size_t)l * stride + i];
        const float2 *j = reinterpret_cast<const float2 *>(dydx) + ((size_t)l * stride + i) * 3;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float2 jj = j[d];
            g[d] = fmaf(ge.x, jj.x, g[d]);
            g[d] = fmaf(ge.y, jj.y, g[d]);
        }
    }
    // d sigma / d h0 from sigma itself: exp(clamp(h0, -80, 80)) (activation.py:20) for sigma = exp(h0); for a softplus density
    // (network.py:115: softplus(h0, beta, threshold 20)) sigmoid(beta h0) = 1 - exp(-beta sigma), 1 in the linear region
    const float sg = sigmas[i];
    const float dact = softplus_beta > 0.0f ? (softplus_beta * sg > 20.0f ? 1.0f : 1.0f - __expf(-softplus_beta * sg))
                                            : fminf(fmaxf(sg, 1.8048513878454153e-35f), 5.5406223843935098e+34f);
    const float k = dact * inv_2bound;
    const float v0 = g[0] * k, v1 = g[1] * k, v2 = g[2] * k;
    const float inv = 1.0f / fmaxf(sqrtf(v0 * v0 + v1 * v1 + v2 * v2), 1e-12f);      // F.normalize
    const float d0 = dirs[(size_t)i * 3], d1 = dirs[(size_t)i * 3 + 1], d2 = dirs[(size_t)i * 3 + 2];
    const float dn = 1.0f / sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float n0 = (-(v0 * inv) + 1.0f) * 0.5f, n1 = (-(v1 * inv) + 1.0f) * 0.5f, n2 = (-(v2 * inv) + 1.0f) * 0.5f;
    const float u0 = d0 * dn, u1 = d1 * dn, u2 = d2 * dn;
    const float ndv = fminf(n0 * -u0 + n1 * -u1 + n2 * -u2, 0.0f);
    term[i] = ndv * ndv;
    if (dterm_ddirs) {   // the normals are constants; through u = d / |d|: (g - u (u . g)) / |d| with g = d term / d u
        const float g0 = -2.0f * ndv * n0, g1 = -2.0f * ndv * n1, g2 = -2.0f * ndv * n2;
        const float dot = g0 * u0 + g1 * u1 + g2 * u2;
        dterm_ddirs[(size_t)i * 3] = (g0 - u0 * dot) * dn;
        dterm_ddirs[(size_t)i * 3 + 1] = (g1 - u1 * dot) * dn;
        dterm_ddirs[(size_t)i * 3 + 2] = (g2 - u2 * dot) * dn;
    }
}

// ------------------------------------------------------------------ d loss / d pose per camera
// rays_o = P[:, 3], rays_d[k] = sum_j dir_cam[j] P[k][j] with dir_cam = ((i + .5 - cx) / fx, -(j + .5 - cy) / fy, -1)
// (train_utils.py:150-160).  One workgroup per camera scans the batch's (view, pixel) list: a fixed summation order.
// grad_pose [V][12] row-major 3 x 4.
__global__ __launch_bounds__(256) void pose_gradient_kernel(const int32_t *__restrict__ index, const float *__restrict__ g_o,
                                                            const float *__restrict__ g_d, uint32_t N, uint32_t W, float fx,
                                                            float fy, float cx, float cy, float *__restrict__ grad_pose)
{
    __shared__ float red[4][12];
    const uint32_t v = blockIdx.x, lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = 0.0f;
    for (uint32_t n = threadIdx.x; n < N; n += 256) {
        if ((uint32_t)index[2 * n] != v) continue;
        const uint32_t pix = (uint32_t)index[2 * n + 1];
        const uint32_t j = pix / W, i = pix - j * W;
        const float dc[3] = {((float)i + 0.5f - cx) / fx, -(((float)j + 0.5f - cy) / fy), -1.0f};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float gd = g_d[(size_t)n * 3 + k];
#pragma unroll
            for (int c = 0; c < 3; c++) acc[4 * k + c] = fmaf(gd, dc[c], acc[4 * k + c]);
            acc[4 * k + 3] += g_o[(size_t)n * 3 + k];
        }
    }
#pragma unroll
    for (int i = 0; i < 12; i++) {
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1) acc[i] += __shfl_xor(acc[i], d, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) red[wid][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < 12) grad_pose[(size_t)v * 12 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// ------------------------------------------------------------------ se(3) exponential with forward-mode derivatives
// value + the 6 partial derivatives with respect to xi = (w, u)
struct Dual6 {
    float v, d[6];
};
__device__ __forceinline__ Dual6 dconst(float c)
{
    Dual6 r;
    r.v = c;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = 0.0f;
    return r;
}
__device__ __forceinline__ Dual6 operator+(const Dual6 &a, const Dual6 &b)
{
    Dual6 r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = a.d[i] + b.d[i];
    return r;
}
__device__ __forceinline__ Dual6 operator-(const Dual6 &a, const Dual6 &b)
{
    Dual6 r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = a.d[i] - b.d[i];
    return r;
}
__device__ __forceinline__ Dual6 operator*(const Dual6 &a, const Dual6 &b)
{
    Dual6 r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
__device__ __forceinline__ Dual6 operator*(float s, const Dual6 &a)
{
    Dual6 r;
    r.v = s * a.v;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = s * a.d[i];
    return r;
}
__device__ __forceinline__ Dual6 dchain(const Dual6 &a, float f, float df)   // f(a) with f' = df
{
    Dual6 r;
    r.v = f;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = df * a.d[i];
    return r;
}

// [exp(w^) | V(w) u] as nerf/pose.py: se3_to_SE3 (barf/camera.py:91-102), series below 1e-2 rad; out[12] row-major 3 x 4
__device__ __forceinline__ void se3_exp(const float (&xi)[6], Dual6 (&out)[12])
{
    Dual6 w[3], u[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        w[i] = dconst(xi[i]);
        w[i].d[i] = 1.0f;
        u[i] = dconst(xi[3 + i]);
        u[i].d[3 + i] = 1.0f;
    }
    const Dual6 t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    // theta = sqrt(clamp_min(|w|^2, 1e-24)): no gradient below the floor, like torch's clamp
    const bool floored = t2.v < 1e-24f;
    const float th = sqrtf(floored ? 1e-24f : t2.v);
    const Dual6 theta = dchain(t2, th, floored ? 0.0f : 0.5f / th);
    Dual6 a, b, c;
    if (th < 1e-2f) {
        const Dual6 q = theta * theta, q2 = q * q;
        a = dconst(1.0f) - (1.0f / 6.0f) * q + (1.0f / 120.0f) * q2;
        b = dconst(0.5f) - (1.0f / 24.0f) * q + (1.0f / 720.0f) * q2;
        c = dconst(1.0f / 6.0f) - (1.0f / 120.0f) * q + (1.0f / 5040.0f) * q2;
    } else {
        const float s = sinf(th), co = cosf(th);
        const Dual6 sn = dchain(theta, s, co), cs = dchain(theta, co, -s);
        const Dual6 inv = dchain(theta, 1.0f / th, -1.0f / (th * th));
        const Dual6 inv2 = inv * inv, inv3 = inv2 * inv;
        a = sn * inv;
        b = (dconst(1.0f) - cs) * inv2;
        c = (theta - sn) * inv3;
    }
    // wx = skew(w), wx2 = wx @ wx
    const Dual6 z = dconst(0.0f);
    const Dual6 wx[9] = {z, z - w[2], w[1], w[2], z, z - w[0], z - w[1], w[0], z};
    Dual6 wx2[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) wx2[3 * r + cc] = wx[3 * r] * wx[cc] + wx[3 * r + 1] * wx[3 + cc] + wx[3 * r + 2] * wx[6 + cc];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        Dual6 tr = z;
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            const Dual6 eye = dconst(r == cc ? 1.0f : 0.0f);
            out[4 * r + cc] = eye + a * wx[3 * r + cc] + b * wx2[3 * r + cc];
            const Dual6 Vrc = eye + b * wx[3 * r + cc] + c * wx2[3 * r + cc];
            tr = tr + Vrc * u[cc];
        }
        out[4 * r + 3] = tr;
    }
}

// refined = compose([exp(xi), base]) (pose.py: compose; barf/camera.py:47-63): R = Rb Rx, t = Rb tx + tb
struct PoseAdam {
    float *exp_avg, *exp_avg_sq;   // [V][6]
    float lr0, gamma, b1, b2, eps;
};

// One thread per camera.  grad_pose == NULL: only (re)compute the refined poses from xi.  Otherwise: gradient of xi through
// the composition, torch.optim.Adam with lr = lr0 * gamma^step (ExponentialLR stepped once per pose step) when flags[0] != 0,
// and the refined pose for the next step.
__global__ void pose_update_kernel(float *__restrict__ xi, const float *__restrict__ base, const float *__restrict__ grad_pose,
                                   uint32_t V, const int32_t *__restrict__ flags, PoseAdam opt, float *__restrict__ refined,
                                   float *__restrict__ grad_xi, const uint32_t *__restrict__ scaler)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float x[6];
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = xi[(size_t)v * 6 + i];
    const float *B = base + (size_t)v * 12;
    if (grad_pose) {
        Dual6 E[12];
        se3_exp(x, E);
        // d loss / d Rx = Rb^T G_R, d loss / d tx = Rb^T G_t
        const float *G = grad_pose + (size_t)v * 12;
        float g[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float up = 0.0f;   // (Rb^T G)[r][c]
#pragma unroll
                for (int k = 0; k < 3; k++) up = fmaf(B[4 * k + r], G[4 * k + c], up);
#pragma unroll
                for (int i = 0; i < 6; i++) g[i] = fmaf(up, E[4 * r + c].d[i], g[i]);
            }
        if (grad_xi) {
#pragma unroll
            for (int i = 0; i < 6; i++) grad_xi[(size_t)v * 6 + i] = g[i];
        }
        // dynamic loss scale (ngp_hip.h; words [2] overflow in this step, [5] steps skipped before it): the reference hands the
        // pose optimiser to the same GradScaler (train_utils.py:898-899) -- no step on overflow, Adam's t counts the steps
        // taken, the ExponentialLR follows the step counter
        const bool skip = scaler && scaler[2] != 0u;
        if (flags[0] != 0 && !skip) {
            const double step = (double)flags[1], t = step + 1.0 - (scaler ? (double)min(scaler[5], (uint32_t)flags[1]) : 0.0);
            const float lr = (float)((double)opt.lr0 * pow((double)opt.gamma, step));
            const float bc1 = (float)(1.0 - pow((double)opt.b1, t)), bc2s = (float)sqrt(1.0 - pow((double)opt.b2, t));
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const size_t k = (size_t)v * 6 + i;
                const float m = opt.b1 * opt.exp_avg[k] + (1.0f - opt.b1) * g[i];
                const float s = opt.b2 * opt.exp_avg_sq[k] + (1.0f - opt.b2) * g[i] * g[i];
                opt.exp_avg[k] = m;
                opt.exp_avg_sq[k] = s;
                x[i] -= (lr / bc1) * (m / (sqrtf(s) / bc2s + opt.eps));
                xi[k] = x[i];
            }
        }
    }
    Dual6 E[12];
    se3_exp(x, E);
    float *P = refined + (size_t)v * 16;
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) P[4 * r + c] = B[4 * r] * E[c].v + B[4 * r + 1] * E[4 + c].v + B[4 * r + 2] * E[8 + c].v;
        P[4 * r + 3] = B[4 * r] * E[3].v + B[4 * r + 1] * E[7].v + B[4 * r + 2] * E[11].v + B[4 * r + 3];
    }
    P[12] = P[13] = P[14] = 0.0f;
    P[15] = 1.0f;
}

}  // namespace ngp

using namespace ngp;

extern "C" int ngp_x_step_window(const uint32_t *step_counter, uint32_t step_offset, double iters, float start_annealing,
                                 float end_annealing, uint32_t L, float *level_w, int32_t *flags, ngp_stream_t stream)
{
    NGP_REQUIRE(step_counter && level_w, "step_window: null tensor");
    NGP_REQUIRE(L >= 1 && L <= 64 && iters > 0.0, "step_window: bad L / iters");
    step_window_kernel<false><<<dim3(1), dim3(64), 0, as_stream(stream)>>>(step_counter, step_offset, iters, start_annealing,
                                                                           end_annealing, L, level_w, flags);
    NGP_CHECK_LAUNCH("step_window");
    return NGP_OK;
}

extern "C" int ngp_x_step_window_baa(const uint32_t *step_counter, uint32_t step_offset, double iters, float start_annealing,
                                     float end_annealing, uint32_t L, float *level_w, int32_t *flags, ngp_stream_t stream)
{
    NGP_REQUIRE(step_counter && level_w, "step_window_baa: null tensor");
    NGP_REQUIRE(L >= 2 && L <= 64 && iters > 0.0, "step_window_baa: bad L / iters");
    step_window_kernel<true><<<dim3(1), dim3(64), 0, as_stream(stream)>>>(step_counter, step_offset, iters, start_annealing,
                                                                          end_annealing, L, level_w, flags);
    NGP_CHECK_LAUNCH("step_window_baa");
    return NGP_OK;
}

extern "C" int ngp_x_slab_window(float *slab, uint32_t stride, uint32_t L, const float *level_w, const int32_t *M_dev,
                                 uint32_t M, int backward, ngp_stream_t stream)
{
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(slab && level_w, "slab_window: null tensor");
    NGP_REQUIRE(L >= 1 && L <= 64 && stride >= M, "slab_window: bad L / stride");
    NGP_REQUIRE(((uintptr_t)slab & 7u) == 0, "slab_window: slab must be 8-byte aligned");
    NGP_REQUIRE(backward >= 0 && backward <= 2, "slab_window: backward must be 0 (blend), 1 (adjoint) or 2 (scale)");
    if (backward == 2)
        slab_window_kernel<2><<<dim3(ceil_div(M, 256u)), dim3(256), 0, as_stream(stream)>>>(
            reinterpret_cast<float2 *>(slab), stride, L, level_w, M_dev, M);
    else if (backward)
        slab_window_kernel<1><<<dim3(ceil_div(M, 256u)), dim3(256), 0, as_stream(stream)>>>(
            reinterpret_cast<float2 *>(slab), stride, L, level_w, M_dev, M);
    else
        slab_window_kernel<0><<<dim3(ceil_div(M, 256u)), dim3(256), 0, as_stream(stream)>>>(
            reinterpret_cast<float2 *>(slab), stride, L, level_w, M_dev, M);
    NGP_CHECK_LAUNCH("slab_window");
    return NGP_OK;
}

extern "C" int ngp_x_ray_gradients(const float *denc, const float *dydx, uint32_t stride, uint32_t L, float bound,
                                   const float *ddirs, const float *ts, const int32_t *rays, uint32_t N, uint32_t M,
                                   float *grad_rays_o, float *grad_rays_d, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays && grad_rays_o && grad_rays_d, "ray_gradients: null tensor");
    NGP_REQUIRE(M == 0 || (denc && dydx && ts), "ray_gradients: null sample tensor");
    NGP_REQUIRE(stride >= M && bound > 0.0f && L >= 1, "ray_gradients: bad stride / bound / L");
    ray_gradients_kernel<<<dim3(ceil_div(N, 4u)), dim3(256), 0, as_stream(stream)>>>(
        denc, dydx, stride, L, 1.0f / (2.0f * bound), ddirs, ts, rays, N, M, grad_rays_o, grad_rays_d, nullptr, nullptr);
    NGP_CHECK_LAUNCH("ray_gradients");
    return NGP_OK;
}

// ... plus a loss term over the samples' weights that depends on the view direction: sd += term_weight[i] * dterm_ddirs[i]
// (both by sample; NULL: none; live_n / live_off NULL: denc in sample order)
extern "C" int ngp_x_ray_gradients_terms(const float *denc, const float *dydx, uint32_t stride, uint32_t L, float bound,
                                         const float *ddirs, const float *ts, const int32_t *rays, const int32_t *live_n,
                                         const int32_t *live_off, const float *term_weight, const float *dterm_ddirs,
                                         uint32_t N, uint32_t M, float *grad_rays_o, float *grad_rays_d, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays && grad_rays_o && grad_rays_d, "ray_gradients_terms: null tensor");
    NGP_REQUIRE(M == 0 || (denc && dydx && ts), "ray_gradients_terms: null sample tensor");
    NGP_REQUIRE(stride >= M && bound > 0.0f && L >= 1, "ray_gradients_terms: bad stride / bound / L");
    NGP_REQUIRE(!live_n == !live_off, "ray_gradients_terms: live_n and live_off go together");
    NGP_REQUIRE(!term_weight == !dterm_ddirs, "ray_gradients_terms: term_weight and dterm_ddirs go together");
    ray_gradients_kernel<<<dim3(ceil_div(N, 4u)), dim3(256), 0, as_stream(stream)>>>(
        denc, dydx, stride, L, 1.0f / (2.0f * bound), ddirs, ts, rays, N, M, grad_rays_o, grad_rays_d, live_n, live_off,
        term_weight, dterm_ddirs);
    NGP_CHECK_LAUNCH("ray_gradients_terms");
    return NGP_OK;
}

// ... when the backward ran over the list of live samples: denc in list order, ray n's entries at live_off[n], live_n[n] of them
extern "C" int ngp_x_ray_gradients_list(const float *denc, const float *dydx, uint32_t stride, uint32_t L, float bound,
                                        const float *ddirs, const float *ts, const int32_t *rays, const int32_t *live_n,
                                        const int32_t *live_off, uint32_t N, uint32_t M, float *grad_rays_o,
                                        float *grad_rays_d, ngp_stream_t stream)
{
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays && grad_rays_o && grad_rays_d && live_n && live_off, "ray_gradients_list: null tensor");
    NGP_REQUIRE(M == 0 || (denc && dydx && ts), "ray_gradients_list: null sample tensor");
    NGP_REQUIRE(stride >= M && bound > 0.0f && L >= 1, "ray_gradients_list: bad stride / bound / L");
    ray_gradients_kernel<<<dim3(ceil_div(N, 4u)), dim3(256), 0, as_stream(stream)>>>(
        denc, dydx, stride, L, 1.0f / (2.0f * bound), ddirs, ts, rays, N, M, grad_rays_o, grad_rays_d, live_n, live_off);
    NGP_CHECK_LAUNCH("ray_gradients_list");
    return NGP_OK;
}

extern "C" int ngp_x_orientation_term(const float *dh_denc, const float *dydx, uint32_t stride, uint32_t L, float bound,
                                      const float *sigmas, const float *dirs, const int32_t *M_dev, uint32_t M, float *term,
                                      float *dterm_ddirs, ngp_stream_t stream)
{
    return ngp_x_orientation_term_act(dh_denc, dydx, stride, L, bound, sigmas, dirs, M_dev, M, term, dterm_ddirs, 0, 1.0f, stream);
}

// ... for a field whose density activation is softplus(beta, threshold 20) (density_act 1; 0: trunc_exp, the entry above)
extern "C" int ngp_x_orientation_term_act(const float *dh_denc, const float *dydx, uint32_t stride, uint32_t L, float bound,
                                          const float *sigmas, const float *dirs, const int32_t *M_dev, uint32_t M, float *term,
                                          float *dterm_ddirs, uint32_t density_act, float beta, ngp_stream_t stream)
{
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(density_act <= 1u && beta > 0.0f, "orientation_term: unknown density activation or beta <= 0");
    NGP_REQUIRE(dh_denc && dydx && sigmas && dirs && term, "orientation_term: null tensor");
    NGP_REQUIRE(stride >= M && bound > 0.0f && L >= 1, "orientation_term: bad stride / bound / L");
    orientation_term_kernel<<<dim3(ceil_div(M, 256u)), dim3(256), 0, as_stream(stream)>>>(
        dh_denc, dydx, stride, L, 1.0f / (2.0f * bound), sigmas, dirs, M_dev, M, term, dterm_ddirs,
        density_act ? beta : 0.0f);
    NGP_CHECK_LAUNCH("orientation_term");
    return NGP_OK;
}

extern "C" int ngp_x_pose_gradient(const int32_t *index, const float *grad_rays_o, const float *grad_rays_d, uint32_t N,
                                   uint32_t V, uint32_t W, float fx, float fy, float cx, float cy, float *grad_pose,
                                   ngp_stream_t stream)
{
    if (V == 0) return NGP_OK;
    NGP_REQUIRE(index && grad_rays_o && grad_rays_d && grad_pose, "pose_gradient: null tensor");
    NGP_REQUIRE(W > 0, "pose_gradient: image width must be positive");
    pose_gradient_kernel<<<dim3(V), dim3(256), 0, as_stream(stream)>>>(index, grad_rays_o, grad_rays_d, N, W, fx, fy, cx, cy,
                                                                       grad_pose);
    NGP_CHECK_LAUNCH("pose_gradient");
    return NGP_OK;
}

extern "C" int ngp_x_pose_update(float *xi, const float *base, const float *grad_pose, uint32_t V, const int32_t *flags,
                                 float *exp_avg, float *exp_avg_sq, float lr0, float gamma, float beta1, float beta2,
                                 float eps, float *refined, float *grad_xi, const float *loss_scaler, ngp_stream_t stream)
{
    if (V == 0) return NGP_OK;
    NGP_REQUIRE(xi && base && refined, "pose_update: null tensor");
    NGP_REQUIRE(!grad_pose || (flags && exp_avg && exp_avg_sq), "pose_update: an update needs flags and the Adam state");
    const PoseAdam opt{exp_avg, exp_avg_sq, lr0, gamma, beta1, beta2, eps};
    pose_update_kernel<<<dim3(ceil_div(V, 64u)), dim3(64), 0, as_stream(stream)>>>(
        xi, base, grad_pose, V, flags, opt, refined, grad_xi, reinterpret_cast<const uint32_t *>(loss_scaler));
    NGP_CHECK_LAUNCH("pose_update");
    return NGP_OK;
}
