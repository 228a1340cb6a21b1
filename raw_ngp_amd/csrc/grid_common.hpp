// Hash-grid geometry shared by the grid-encoder kernels (forward, atomic backward, binned backward).
// Index rule and cell location follow gridencoder/src/gridencoder.cu:45-79 and :105-160 of the reference;
// arithmetic mirrors oracle/ngp_oracle.c (explicit fmaf, no implicit contraction).
#pragma once
#include "ngp_common.hpp"

namespace ngp {

constexpr uint32_t kMaxLevels = 64;
constexpr uint32_t kBlock = 256;

struct LevelRes {
    uint32_t res[kMaxLevels];
};

__device__ constexpr uint32_t kPrimes[7] = {1u,          2654435761u, 805459861u, 3674653429u,
                                            2097192037u, 1434869437u, 2165219737u};

// Per-level geometry; every field is workgroup-uniform (lives in SGPRs).
template <uint32_t D>
struct Geom {
    uint32_t T;          // rows in this level
    uint32_t res;
    uint32_t stride[D];  // dense strides (uint32 wrap like the reference)
    uint32_t nd;         // dims that entered the dense index before stride exceeded T
    bool hashed;
    uint32_t mode;       // 0: index < T by construction, 1: T is a power of two (mask), 2: modulo
};

template <uint32_t D>
__device__ __forceinline__ Geom<D> make_geom(const int32_t *__restrict__ offsets, uint32_t level, uint32_t res,
                                            uint32_t gridtype)
{
    Geom<D> g;
    g.T = (uint32_t)(offsets[level + 1] - offsets[level]);
    g.res = res;
    uint32_t stride = 1, d = 0;
#pragma unroll
    for (uint32_t k = 0; k < D; k++) g.stride[k] = 0;
#pragma unroll
    for (uint32_t k = 0; k < D; k++) {
        if (d == k && stride <= g.T) {
            g.stride[k] = stride;
            stride *= res;
            d = k + 1;
        }
    }
    g.nd = d;
    g.hashed = (gridtype == 0u) && (stride > g.T);
    if (!g.hashed && d == D && stride <= g.T)
        g.mode = 0;
    else if ((g.T & (g.T - 1u)) == 0u)
        g.mode = 1;
    else
        g.mode = 2;
    return g;
}

template <uint32_t D>
__device__ __forceinline__ uint32_t row_of(const Geom<D> &g, const uint32_t (&c)[D])
{
    uint32_t idx = 0;
    if (g.hashed) {
#pragma unroll
        for (uint32_t d = 0; d < D; d++) idx ^= c[d] * kPrimes[d];
    } else {
#pragma unroll
        for (uint32_t d = 0; d < D; d++) idx += c[d] * g.stride[d];  // stride 0 for dims past nd
    }
    if (g.mode == 1)
        idx &= g.T - 1u;
    else if (g.mode == 2)
        idx %= g.T;
    return idx;
}

// Per-axis index terms of a cell's two coordinates (c and min(c + 1, res - 1)): c * prime (hashed) or c * stride
// (dense).  The 2^D corner rows are combinations of these, so the multiplies are done D times instead of D * 2^D
// (v_mul_lo_u32 runs at quarter rate) and the second term of an axis is the first plus the multiplier.
template <uint32_t D>
struct AxisTerms {
    uint32_t t[D][2];
};

template <uint32_t D>
__device__ __forceinline__ uint32_t row_from_terms(const Geom<D> &g, const AxisTerms<D> &a, uint32_t corner)
{
    uint32_t idx = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t v = a.t[d][(corner >> d) & 1u];
        idx = g.hashed ? (idx ^ v) : (idx + v);
    }
    if (g.mode == 1)
        idx &= g.T - 1u;
    else if (g.mode == 2)
        idx %= g.T;
    return idx;
}

template <uint32_t D>
struct Cell {
    uint32_t c[D];
    float f[D];
    float df[D];
};

template <uint32_t D>
__device__ __forceinline__ AxisTerms<D> axis_terms(const Geom<D> &g, const Cell<D> &cl)
{
    AxisTerms<D> a;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t m = g.hashed ? kPrimes[d] : g.stride[d];
        a.t[d][0] = cl.c[d] * m;
        a.t[d][1] = cl.c[d] + 1u <= g.res - 1u ? a.t[d][0] + m : a.t[d][0];   // c + 1 clamped to res - 1
    }
    return a;
}

// false when the point is outside [0,1]^D (the reference zeroes / skips those)
template <uint32_t D>
__device__ __forceinline__ bool locate(const float (&x)[D], uint32_t res, bool align_corners, uint32_t interp,
                                       Cell<D> &o)
{
    bool inside = true;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) inside = inside && !(x[d] < 0.0f || x[d] > 1.0f);
    if (!inside) return false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        float p;
        uint32_t c;
        if (align_corners) {
            p = x[d] * (float)(res - 1u);
            c = min((uint32_t)floorf(p), res - 2u);
        } else {
            p = fminf(fmaxf(fmaf(x[d], (float)res, -0.5f), 0.0f), (float)(res - 1u));
            c = (uint32_t)floorf(p);
        }
        p -= (float)c;
        if (interp == 1u) {
            o.df[d] = 6.0f * p * (1.0f - p);
            p = p * p * (3.0f - 2.0f * p);
        } else {
            o.df[d] = 1.0f;
        }
        o.c[d] = c;
        o.f[d] = p;
    }
    return true;
}

// Row access in the widest vector the channel count allows.
template <uint32_t C>
struct Row {
    float v[C];
    __device__ __forceinline__ void load(const float *__restrict__ p)
    {
        if constexpr (C == 1) {
            v[0] = p[0];
        } else if constexpr (C == 2) {
            const float2 t = *reinterpret_cast<const float2 *>(p);
            v[0] = t.x;
            v[1] = t.y;
        } else {
#pragma unroll
            for (uint32_t i = 0; i < C; i += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(p + i);
                v[i] = t.x;
                v[i + 1] = t.y;
                v[i + 2] = t.z;
                v[i + 3] = t.w;
            }
        }
    }
    __device__ __forceinline__ void store_stream(float *__restrict__ p) const
    {
        if constexpr (C == 1) {
            __builtin_nontemporal_store(v[0], p);
        } else if constexpr (C == 2) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 t = {v[0], v[1]};
            __builtin_nontemporal_store(t, reinterpret_cast<f2 *>(p));
        } else {
            typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (uint32_t i = 0; i < C; i += 4) {
                f4 t = {v[i], v[i + 1], v[i + 2], v[i + 3]};
                __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(p + i));
            }
        }
    }
};

// per-level resolutions, evaluated in float32 on the host exactly like oracle/ngp_oracle.c
static inline bool fill_levels(LevelRes &lv, float S, uint32_t H, uint32_t L)
{
    if (L == 0 || L > kMaxLevels) return false;
    for (uint32_t l = 0; l < L; l++) lv.res[l] = (uint32_t)ceilf(exp2f((float)l * S) * (float)H);
    for (uint32_t l = L; l < kMaxLevels; l++) lv.res[l] = 0;
    return true;
}

}  // namespace ngp
