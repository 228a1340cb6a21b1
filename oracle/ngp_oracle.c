/*
 * ngp_oracle.c -- CPU restatement of raw_ngp's data-parallel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package (raw_ngp_amd/) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker / the timed CPU
 * baseline.  The product path is the HIP library behind include/ngp_hip.h.
 *
 * What it restates (all file:line into the reference tree):
 *   grid encoder    gridencoder/src/gridencoder.cu:45-79 (hash / index),
 *                   :82-249 (forward + dy_dx), :252-349 (table gradient),
 *                   :352-378 (input gradient), :525-631 (TV), :670-703 (WD)
 *   SH encoder      shencoder/src/shencoder.cu:43-121 (values), :130-350
 *                   (ambient Jacobian), :358-382 (backward)
 *   ray marching    raymarching/src/raymarching.cu:42-81 (mip, morton),
 *                   :91-145, :162-198, :214-254, :267-289, :303-319,
 *                   :337-491 (train march), :519-597, :623-712 (compositing),
 *                   :731-846, :860-941 (inference pair)
 *   freq encoder    freqencoder/src/freqencoder.cu:30-94
 *
 * Pinning status: the reference ships no tests, golden vectors or fixtures for
 * these kernels and its CUDA sources cannot be built in this
 * image (no nvcc), so the kernels are pinned by first-principles known-answer
 * tests (tests/test_oracle_*.py: grid_sample, scipy sph_harm, numpy packbits,
 * closed-form march counts, cumsum compositing) and, for the Python-side pieces
 * that do run on CPU, by fixtures generated from the reference's own Python
 * (oracle/gen_golden.py -> tests/golden/).  Versus the real CUDA binaries the
 * parity of these kernels is therefore "unpinned"; see DESIGN.md.
 *
 * Float semantics: compiled with -ffp-contract=off.  Where nvcc's default
 * -fmad=true would fuse a multiply feeding an add inside one expression, fmaf()
 * is written explicitly; the HIP kernels make the same choices so that integer
 * results (cells, sample counts) agree bit for bit between the two.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_D 5
#define ORC_MAX_C 32

int orc_version(void) { return 1; }

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static inline float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

/* ------------------------------------------------------------------ */
/* grid encoder                                                         */
/* ------------------------------------------------------------------ */

static const uint32_t k_hash_primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                          2097192037u, 1434869437u, 2165219737u};

/* gridencoder.cu:133 -- per-level resolution, evaluated in float32 on the host */
void orc_grid_resolutions(float S, uint32_t H, uint32_t L, uint32_t *res)
{
    for (uint32_t l = 0; l < L; l++)
        res[l] = (uint32_t)ceilf(exp2f((float)l * S) * (float)H);
}

/* gridencoder.cu:61-79 -- row index (without the *C + ch part) */
static inline uint32_t grid_row(uint32_t gridtype, uint32_t D, uint32_t T, uint32_t res,
                                const uint32_t *c)
{
    uint32_t stride = 1, row = 0, d;
    for (d = 0; d < D && stride <= T; d++) {
        row += c[d] * stride;
        stride *= res;
    }
    if (gridtype == 0 && stride > T) {
        row = 0;
        for (d = 0; d < D; d++) row ^= c[d] * k_hash_primes[d];
    }
    return row % T;
}

typedef struct {
    uint32_t cell[ORC_MAX_D];
    float frac[ORC_MAX_D];  /* interpolation weight toward cell+1 */
    float dfrac[ORC_MAX_D]; /* d(frac)/d(pos) */
} cell_t;

/* returns 0 when the point is outside [0,1]^D (gridencoder.cu:105-112) */
static inline int grid_locate(const float *x, uint32_t D, uint32_t res, int align_corners,
                              uint32_t interp, cell_t *o)
{
    uint32_t d;
    for (d = 0; d < D; d++)
        if (x[d] < 0.0f || x[d] > 1.0f) return 0;
    for (d = 0; d < D; d++) {
        float p;
        uint32_t c;
        if (align_corners) {
            p = x[d] * (float)(res - 1);
            c = (uint32_t)floorf(p);
            if (c > res - 2) c = res - 2;
        } else {
            p = fminf(fmaxf(fmaf(x[d], (float)res, -0.5f), 0.0f), (float)(res - 1));
            c = (uint32_t)floorf(p);
        }
        p -= (float)c;
        if (interp == 1) {
            o->dfrac[d] = 6.0f * p * (1.0f - p);
            p = p * p * (3.0f - 2.0f * p);
        } else {
            o->dfrac[d] = 1.0f;
        }
        o->cell[d] = c;
        o->frac[d] = p;
    }
    return 1;
}

void orc_grid_encode_forward(const float *inputs, const float *table, const int32_t *offsets,
                             float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                             uint32_t max_level, float S, uint32_t H, float *dy_dx,
                             uint32_t gridtype, int align_corners, uint32_t interp)
{
    uint32_t res_tab[64];
    orc_grid_resolutions(S, H, L, res_tab);
    for (uint32_t level = 0; level < max_level; level++) {
        const float *tab = table + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t T = (uint32_t)(offsets[level + 1] - offsets[level]);
        const uint32_t res = res_tab[level];
        const float scale = (float)(align_corners ? res - 1 : res);
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            const float *x = inputs + (size_t)b * D;
            float *out = outputs + ((size_t)level * B + (size_t)b) * C;
            float *jac = dy_dx ? dy_dx + (size_t)b * L * D * C + (size_t)level * D * C : NULL;
            cell_t cl;
            uint32_t ch, d;
            if (!grid_locate(x, D, res, align_corners, interp, &cl)) {
                for (ch = 0; ch < C; ch++) out[ch] = 0.0f;
                if (jac)
                    for (d = 0; d < D * C; d++) jac[d] = 0.0f;
                continue;
            }
            float acc[ORC_MAX_C];
            for (ch = 0; ch < C; ch++) acc[ch] = 0.0f;
            for (uint32_t corner = 0; corner < (1u << D); corner++) {
                float w = 1.0f;
                uint32_t c[ORC_MAX_D];
                for (d = 0; d < D; d++) {
                    if (corner & (1u << d)) {
                        w *= cl.frac[d];
                        c[d] = cl.cell[d] + 1 < res - 1 ? cl.cell[d] + 1 : res - 1;
                    } else {
                        w *= 1.0f - cl.frac[d];
                        c[d] = cl.cell[d];
                    }
                }
                const float *row = tab + (size_t)grid_row(gridtype, D, T, res, c) * C;
                for (ch = 0; ch < C; ch++) acc[ch] = fmaf(w, row[ch], acc[ch]);
            }
            for (ch = 0; ch < C; ch++) out[ch] = acc[ch];

            if (!jac) continue;
            /* gridencoder.cu:205-247 */
            for (uint32_t gd = 0; gd < D; gd++) {
                float g[ORC_MAX_C];
                for (ch = 0; ch < C; ch++) g[ch] = 0.0f;
                for (uint32_t combo = 0; combo < (1u << (D - 1)); combo++) {
                    float w = scale;
                    uint32_t c[ORC_MAX_D];
                    for (uint32_t nd = 0; nd < D - 1; nd++) {
                        d = nd >= gd ? nd + 1 : nd;
                        if (combo & (1u << nd)) {
                            w *= cl.frac[d];
                            c[d] = cl.cell[d] + 1 < res - 1 ? cl.cell[d] + 1 : res - 1;
                        } else {
                            w *= 1.0f - cl.frac[d];
                            c[d] = cl.cell[d];
                        }
                    }
                    c[gd] = cl.cell[gd];
                    const float *lo = tab + (size_t)grid_row(gridtype, D, T, res, c) * C;
                    c[gd] = cl.cell[gd] + 1 < res - 1 ? cl.cell[gd] + 1 : res - 1;
                    const float *hi = tab + (size_t)grid_row(gridtype, D, T, res, c) * C;
                    for (ch = 0; ch < C; ch++)
                        g[ch] = fmaf(w * (hi[ch] - lo[ch]), cl.dfrac[gd], g[ch]);
                }
                for (ch = 0; ch < C; ch++) jac[gd * C + ch] = g[ch];
            }
        }
    }
}

/* gridencoder.cu:252-349.  The reference adds with float atomics in an order the
 * hardware picks; here every table entry is summed in double in (level, b, corner)
 * order and rounded once, i.e. the centre of all orders the reference can produce.
 * grad_table must be pre-zeroed by the caller (grid.py:83) -- we ADD into it. */
void orc_grid_encode_backward(const float *grad, const float *inputs, const float *table,
                              const int32_t *offsets, float *grad_table, uint32_t B, uint32_t D,
                              uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                              const float *dy_dx, float *grad_inputs, uint32_t gridtype,
                              int align_corners, uint32_t interp)
{
    (void)table;
    uint32_t res_tab[64];
    orc_grid_resolutions(S, H, L, res_tab);
    const size_t n_rows = (size_t)(uint32_t)offsets[L];
    double *acc = (double *)calloc(n_rows * C, sizeof(double));
    /* levels own disjoint row ranges: parallel over levels keeps the per-row order (level, b, corner) */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t level = 0; level < (int64_t)max_level; level++) {
        double *gt = acc + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t T = (uint32_t)(offsets[level + 1] - offsets[level]);
        const uint32_t res = res_tab[level];
        for (uint32_t b = 0; b < B; b++) {
            const float *x = inputs + (size_t)b * D;
            const float *g = grad + ((size_t)level * B + b) * C;
            cell_t cl;
            if (!grid_locate(x, D, res, align_corners, interp, &cl)) continue;
            for (uint32_t corner = 0; corner < (1u << D); corner++) {
                float w = 1.0f;
                uint32_t c[ORC_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if (corner & (1u << d)) {
                        w *= cl.frac[d];
                        c[d] = cl.cell[d] + 1 < res - 1 ? cl.cell[d] + 1 : res - 1;
                    } else {
                        w *= 1.0f - cl.frac[d];
                        c[d] = cl.cell[d];
                    }
                }
                double *row = gt + (size_t)grid_row(gridtype, D, T, res, c) * C;
                for (uint32_t ch = 0; ch < C; ch++) row[ch] += (double)(w * g[ch]);
            }
        }
    }
    for (size_t i = 0; i < n_rows * C; i++) grad_table[i] += (float)acc[i];
    free(acc);

    /* gridencoder.cu:352-378 -- all L levels, also when max_level < L */
    if (dy_dx && grad_inputs) {
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)B * D; t++) {
            const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
            const float *jac = dy_dx + (size_t)b * L * D * C;
            float r = 0.0f;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++)
                    r = fmaf(grad[((size_t)l * B + b) * C + ch], jac[(size_t)l * D * C + d * C + ch], r);
            grad_inputs[t] = r;
        }
    }
}

/* gridencoder.cu:525-631.  Adds into grad (double accumulation, rounded once). */
void orc_grad_total_variation(const float *inputs, const float *table, float *grad,
                              const int32_t *offsets, float weight, uint32_t B, uint32_t D,
                              uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                              int align_corners)
{
    uint32_t res_tab[64];
    orc_grid_resolutions(S, H, L, res_tab);
    const size_t n_rows = (size_t)(uint32_t)offsets[L];
    double *acc = (double *)calloc(n_rows * C, sizeof(double));
    const float w = weight / (float)(2 * D);
    for (uint32_t level = 0; level < L; level++) {
        const float *tab = table + (size_t)(uint32_t)offsets[level] * C;
        double *ga = acc + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t T = (uint32_t)(offsets[level + 1] - offsets[level]);
        const uint32_t res = res_tab[level];
        for (uint32_t b = 0; b < B; b++) {
            const float *x = inputs + (size_t)b * D;
            cell_t cl;
            /* TV never uses smoothstep; only the cell matters */
            if (!grid_locate(x, D, res, align_corners, 0, &cl)) continue;
            uint32_t c[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) c[d] = cl.cell[d];
            const uint32_t centre = grid_row(gridtype, D, T, res, c);
            float sum[ORC_MAX_C], sq[ORC_MAX_C];
            for (uint32_t ch = 0; ch < C; ch++) sum[ch] = sq[ch] = 0.0f;
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t cur = c[d];
                /* + side: the reference's guard `cur < resolution` is always true, so the
                 * neighbour may sit at index == resolution (not clamped) -- :595 */
                if (cur < res) {
                    c[d] = cur + 1;
                    const float *nb = tab + (size_t)grid_row(gridtype, D, T, res, c) * C;
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float dv = tab[(size_t)centre * C + ch] - nb[ch];
                        sum[ch] += dv;
                        sq[ch] = fmaf(dv, dv, sq[ch]);
                    }
                }
                if (cur > 0) {
                    c[d] = cur - 1;
                    const float *nb = tab + (size_t)grid_row(gridtype, D, T, res, c) * C;
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float dv = tab[(size_t)centre * C + ch] - nb[ch];
                        sum[ch] += dv;
                        sq[ch] = fmaf(dv, dv, sq[ch]);
                    }
                }
                c[d] = cur;
            }
            for (uint32_t ch = 0; ch < C; ch++)
                ga[(size_t)centre * C + ch] += (double)(w * sum[ch] * (1.0f / sqrtf(sq[ch] + 1e-9f)));
        }
    }
    for (size_t i = 0; i < n_rows * C; i++) grad[i] += (float)acc[i];
    free(acc);
}

/* gridencoder.cu:670-703 */
void orc_grad_weight_decay(const float *table, float *grad, const int32_t *offsets, float weight,
                           uint32_t B, uint32_t C, uint32_t L)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)B * C; i++) {
        const uint32_t n = (uint32_t)(i / C);
        uint32_t level = 0, lo = 0, hi = L;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
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

/* ------------------------------------------------------------------ */
/* spherical harmonics                                                  */
/* ------------------------------------------------------------------ */
/* The reference hard-codes, for degree <= 8, the real SH basis in Cartesian form
 *   Y_{l,m}(x,y,z) = s_m K_{l,|m|} Q_l^{|m|}(z) * { Im (x+iy)^{|m|}  (m<0) | 1 (m=0) | Re (x+iy)^m (m>0) }
 * with Q_l^m(z) = d^m/dz^m P_l(z) (a polynomial in z only), s_m = (-1)^m and
 * K = sqrt((2l+1)/(4 pi) (l-|m|)!/(l+|m|)!) * (m ? sqrt 2 : 1); output index l*l+l+m
 * (shencoder.cu:50-120: Y_1 = 0.4886 (-y, z, -x), Y_2,0 = 0.9462 z^2 - 0.3154, ...).
 * Its Jacobian (shencoder.cu:130-350) is the ambient derivative of exactly those
 * polynomial representatives (NOT projected to the sphere), so the representative
 * matters.  We build the monomial table from that closed form at first use and
 * evaluate values and derivatives from the table in double precision. */

#define SH_MAX_DEG 8
#define SH_MAX_OUT 64
#define SH_MAX_TERMS 40

typedef struct {
    int n;
    double coef[SH_MAX_TERMS];
    unsigned char ex[SH_MAX_TERMS], ey[SH_MAX_TERMS], ez[SH_MAX_TERMS];
} sh_poly_t;

static sh_poly_t g_sh[SH_MAX_OUT];
static int g_sh_ready = 0;

static double dfact(int n)
{
    double r = 1.0;
    for (int i = 2; i <= n; i++) r *= i;
    return r;
}

static double binom(int n, int k) { return dfact(n) / (dfact(k) * dfact(n - k)); }

static void sh_build(void)
{
    double P[SH_MAX_DEG][SH_MAX_DEG]; /* P[l][k]: coefficient of z^k in P_l */
    memset(P, 0, sizeof(P));
    P[0][0] = 1.0;
    P[1][1] = 1.0;
    for (int l = 1; l + 1 < SH_MAX_DEG; l++)
        for (int k = 0; k < SH_MAX_DEG; k++) {
            double v = -(double)l * P[l - 1][k];
            if (k > 0) v += (2.0 * l + 1.0) * P[l][k - 1];
            P[l + 1][k] = v / (l + 1.0);
        }
    for (int l = 0; l < SH_MAX_DEG; l++)
        for (int m = -l; m <= l; m++) {
            const int am = m < 0 ? -m : m;
            sh_poly_t *p = &g_sh[l * l + l + m];
            p->n = 0;
            double K = sqrt((2.0 * l + 1.0) / (4.0 * M_PI) * dfact(l - am) / dfact(l + am));
            if (am) K *= sqrt(2.0) * ((am & 1) ? -1.0 : 1.0);
            /* Q(z) = d^am/dz^am P_l */
            double Q[SH_MAX_DEG];
            memset(Q, 0, sizeof(Q));
            for (int k = am; k < SH_MAX_DEG; k++) {
                double f = 1.0;
                for (int j = 0; j < am; j++) f *= (k - j);
                Q[k - am] = P[l][k] * f;
            }
            /* angular part: Re or Im of (x + i y)^am */
            for (int k = 0; k <= am; k++) {
                const int is_im = k & 1;
                if (am == 0 ? k != 0 : (m > 0 ? is_im : !is_im)) continue;
                const double sgn = ((k / 2) & 1) ? -1.0 : 1.0;
                const double a = binom(am, k) * sgn;
                for (int kz = 0; kz < SH_MAX_DEG; kz++) {
                    if (Q[kz] == 0.0) continue;
                    p->coef[p->n] = K * a * Q[kz];
                    p->ex[p->n] = (unsigned char)(am - k);
                    p->ey[p->n] = (unsigned char)k;
                    p->ez[p->n] = (unsigned char)kz;
                    p->n++;
                }
            }
        }
    g_sh_ready = 1;
}

static inline double ipow(double v, int e)
{
    double r = 1.0;
    while (e-- > 0) r *= v;
    return r;
}

void orc_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D,
                           uint32_t degree, float *dy_dx)
{
    if (!g_sh_ready) {
#pragma omp critical
        if (!g_sh_ready) sh_build();
    }
    const uint32_t n_out = degree * degree;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        const double x = inputs[(size_t)b * D], y = inputs[(size_t)b * D + 1], z = inputs[(size_t)b * D + 2];
        float *out = outputs + (size_t)b * n_out;
        float *jac = dy_dx ? dy_dx + (size_t)b * D * n_out : NULL;
        for (uint32_t i = 0; i < n_out; i++) {
            const sh_poly_t *p = &g_sh[i];
            double v = 0, gx = 0, gy = 0, gz = 0;
            for (int t = 0; t < p->n; t++) {
                const int ex = p->ex[t], ey = p->ey[t], ez = p->ez[t];
                const double c = p->coef[t];
                v += c * ipow(x, ex) * ipow(y, ey) * ipow(z, ez);
                if (jac) {
                    if (ex) gx += c * ex * ipow(x, ex - 1) * ipow(y, ey) * ipow(z, ez);
                    if (ey) gy += c * ey * ipow(x, ex) * ipow(y, ey - 1) * ipow(z, ez);
                    if (ez) gz += c * ez * ipow(x, ex) * ipow(y, ey) * ipow(z, ez - 1);
                }
            }
            out[i] = (float)v;
            if (jac) {
                jac[i] = (float)gx;
                jac[n_out + i] = (float)gy;
                jac[2 * n_out + i] = (float)gz;
            }
        }
    }
}

/* shencoder.cu:358-382 -- accumulates into grad_inputs (caller zero-inits) */
void orc_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D,
                            uint32_t degree, const float *dy_dx, float *grad_inputs)
{
    (void)inputs;
    const uint32_t n_out = degree * degree;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * D; t++) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
        const float *g = grad + (size_t)b * n_out;
        const float *j = dy_dx + (size_t)b * D * n_out + (size_t)d * n_out;
        float r = grad_inputs[t];
        for (uint32_t ch = 0; ch < n_out; ch++) r = fmaf(g[ch], j[ch], r);
        grad_inputs[t] = r;
    }
}

/* ------------------------------------------------------------------ */
/* frequency encoder (freqencoder.cu:30-94)                             */
/* ------------------------------------------------------------------ */
/* outputs [B, C] with C = D + D*deg*2: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] per
 * frequency block of D*2 ... laid out as in the reference: out[b, D + f*2*D + k] with
 * k<D -> sin(2^f x_k), k>=D -> cos(2^f x_{k-D}).  cos is evaluated as sin(. + pi/2). */
void orc_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                             float *outputs)
{
    const float half_pi = 1.57079632679489661923f;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * C; t++) {
        const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (int64_t)b * C);
        const float *x = inputs + (size_t)b * D;
        if (c < D) {
            outputs[t] = x[c];
        } else {
            const uint32_t col = c / D - 1, d = c % D;
            const uint32_t freq = col / 2;
            const float phase = (float)(col % 2) * half_pi;
            outputs[t] = sinf(scalbnf(x[d], (int)freq) + phase);
        }
    }
}

/* grad_inputs[b,d] = g[b,d] + sum_f 2^f (g_sin * cos_out - g_cos * sin_out) */
void orc_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D,
                              uint32_t deg, uint32_t C, float *grad_inputs)
{
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * D; t++) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
        const float *g = grad + (size_t)b * C;
        const float *o = outputs + (size_t)b * C;
        float r = g[d];
        g += D;
        o += D;
        for (uint32_t f = 0; f < deg; f++) {
            r += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
            g += 2 * D;
            o += 2 * D;
        }
        grad_inputs[t] = r;
    }
}

/* ------------------------------------------------------------------ */
/* ray marching: helpers                                                */
/* ------------------------------------------------------------------ */

#define ORC_SQRT3 1.7320508075688772f
#define ORC_RPI 0.3183098861837907f

static inline uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z)
{
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}

static inline uint32_t compact_bits(uint32_t x)
{
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* raymarching.cu:42-54: frexp exponent, clamped to [0, cascades-1] */
static inline int mip_of(float mx, float cascades)
{
    int e;
    frexpf(mx, &e);
    return (int)fminf(cascades - 1.0f, fmaxf(0.0f, (float)e));
}

void orc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                            float min_near, float *nears, float *fars)
{
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float *o = rays_o + 3 * n, *d = rays_d + 3 * n;
        float tn = -FLT_MAX, tf = FLT_MAX;
        int miss = 0;
        /* slab by slab in x, y, z order; the miss test uses the running interval
         * exactly like raymarching.cu:113-139 */
        for (int a = 0; a < 3 && !miss; a++) {
            const float r = 1.0f / d[a];
            float t0 = (aabb[a] - o[a]) * r, t1 = (aabb[3 + a] - o[a]) * r;
            if (t0 > t1) {
                const float s = t0;
                t0 = t1;
                t1 = s;
            }
            if (a == 0) {
                tn = t0;
                tf = t1;
            } else {
                if (tn > t1 || t0 > tf) {
                    miss = 1;
                    break;
                }
                if (t0 > tn) tn = t0;
                if (t1 < tf) tf = t1;
            }
        }
        if (miss) {
            nears[n] = fars[n] = FLT_MAX;
            continue;
        }
        if (tn < min_near) tn = min_near;
        nears[n] = tn;
        fars[n] = tf;
    }
}

void orc_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords)
{
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float *o = rays_o + 3 * n, *d = rays_d + 3 * n;
        const float A = fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0]));
        const float Bh = fmaf(o[2], d[2], fmaf(o[1], d[1], o[0] * d[0]));
        const float Cc = fmaf(o[2], o[2], fmaf(o[1], o[1], o[0] * o[0])) - radius * radius;
        const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;
        const float x = fmaf(t, d[0], o[0]), y = fmaf(t, d[1], o[1]), z = fmaf(t, d[2], o[2]);
        const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
        const float phi = atan2f(z, x);
        coords[2 * n] = 2.0f * theta * ORC_RPI - 1.0f;
        coords[2 * n + 1] = phi * ORC_RPI;
    }
}

void orc_morton3D(const int32_t *coords, uint32_t N, int32_t *indices)
{
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)morton3((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1],
                                      (uint32_t)coords[3 * n + 2]);
}

void orc_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords)
{
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t v = (uint32_t)indices[n];
        coords[3 * n] = (int32_t)compact_bits(v);
        coords[3 * n + 1] = (int32_t)compact_bits(v >> 1);
        coords[3 * n + 2] = (int32_t)compact_bits(v >> 2);
    }
}

void orc_packbits(const float *grid, uint32_t N, float thresh, uint8_t *bitfield)
{
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++)
            if (grid[8 * n + i] > thresh) bits |= (uint8_t)(1u << i);
        bitfield[n] = bits;
    }
}

void orc_flatten_rays(const int32_t *rays, uint32_t N, uint32_t M, int32_t *res)
{
    (void)M;
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
        for (uint32_t i = 0; i < cnt; i++) res[off + i] = (int32_t)n;
    }
}

/* ------------------------------------------------------------------ */
/* ray marching: the stepping rule shared by training and inference      */
/* ------------------------------------------------------------------ */

typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Hf, cascades;
    int contract;
    uint32_t H;
    const uint8_t *grid;
} march_t;

/* One evaluation of the marcher at parameter t (raymarching.cu:407-480).
 * Returns 1 when a sample is emitted (p = contracted position, *dt = its step and the
 * caller advances t by *dt), 0 when empty space was skipped (t already advanced). */
static inline int march_probe(const march_t *m, float *t, float *dt_out, float p[3])
{
    const float x = clampf(fmaf(*t, m->dx, m->ox), -m->bound, m->bound);
    const float y = clampf(fmaf(*t, m->dy, m->oy), -m->bound, m->bound);
    const float z = clampf(fmaf(*t, m->dz, m->oz), -m->bound, m->bound);
    float dt = clampf(*t * m->dt_gamma, m->dt_min, m->dt_max);

    const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    const int lvl_p = mip_of(mag, m->cascades);
    const int lvl_d = mip_of(dt * m->Hf * 0.5f, m->cascades);
    const int level = lvl_p > lvl_d ? lvl_p : lvl_d;
    const float mip_bound = fminf(scalbnf(1.0f, level), m->bound);
    const float mip_rbound = 1.0f / mip_bound;

    float cx = x, cy = y, cz = z;
    const int outside = m->contract && mag > 1.0f;
    if (outside) {
        const float s = (2.0f - 1.0f / mag) / mag;
        cx *= s;
        cy *= s;
        cz *= s;
    }
    /* 0.5 * (c / mip_bound + 1) * H, truncated toward zero after clamping.  The
     * reference evaluates the product in double (literal 0.5) and narrows; scaling
     * by 0.5 is exact, so one float multiply by H rounds identically. */
    const int nx = (int)clampf(0.5f * fmaf(cx, mip_rbound, 1.0f) * m->Hf, 0.0f, (float)(m->H - 1));
    const int ny = (int)clampf(0.5f * fmaf(cy, mip_rbound, 1.0f) * m->Hf, 0.0f, (float)(m->H - 1));
    const int nz = (int)clampf(0.5f * fmaf(cz, mip_rbound, 1.0f) * m->Hf, 0.0f, (float)(m->H - 1));

    const uint32_t bit = (uint32_t)((float)level * m->H3 + (float)morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = (m->grid[bit / 8] >> (bit % 8)) & 1;

    if (occ || outside) {
        p[0] = cx;
        p[1] = cy;
        p[2] = cz;
        *dt_out = dt;
        return 1;
    }
    const float sx = copysignf(1.0f, m->dx), sy = copysignf(1.0f, m->dy), sz = copysignf(1.0f, m->dz);
    const float tx = fmaf(fmaf(((float)nx + 0.5f + 0.5f * sx) * m->rH, 2.0f, -1.0f), mip_bound, -cx) * m->rdx;
    const float ty = fmaf(fmaf(((float)ny + 0.5f + 0.5f * sy) * m->rH, 2.0f, -1.0f), mip_bound, -cy) * m->rdy;
    const float tz = fmaf(fmaf(((float)nz + 0.5f + 0.5f * sz) * m->rH, 2.0f, -1.0f), mip_bound, -cz) * m->rdz;
    const float tt = *t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        dt = clampf(*t * m->dt_gamma, m->dt_min, m->dt_max);
        *t += dt;
    } while (*t < tt);
    return 0;
}

static inline void march_setup(march_t *m, const float *o, const float *d, int inference, float bound,
                               int contract, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                               const uint8_t *grid)
{
    m->ox = o[0];
    m->oy = o[1];
    m->oz = o[2];
    m->dx = d[0];
    m->dy = d[1];
    m->dz = d[2];
    if (inference) { /* raymarching.cu:762 */
        m->rdx = 1.0f / (d[0] + 1e-10f);
        m->rdy = 1.0f / (d[1] + 1e-10f);
        m->rdz = 1.0f / (d[2] + 1e-10f);
    } else { /* :388 */
        m->rdx = 1.0f / d[0];
        m->rdy = 1.0f / d[1];
        m->rdz = 1.0f / d[2];
    }
    m->bound = bound;
    m->contract = contract;
    m->dt_gamma = dt_gamma;
    m->dt_min = 2.0f * ORC_SQRT3 / (float)max_steps;
    m->dt_max = 2.0f * ORC_SQRT3 * bound / (float)H;
    m->H = H;
    m->Hf = (float)H;
    m->rH = 1.0f / (float)H;
    m->H3 = (float)(H * H * H);
    m->cascades = (float)C;
    m->grid = grid;
}

/* Training march, both passes of raymarching.cu:337-491 in one call:
 *   pass A counts samples per ray; offsets are the exclusive prefix sum in ray order
 *   (one of the orders the reference's atomicAdd can produce, and the one its own
 *   backward assumes -- raymarching.py:325-328); counter[0] += total.
 *   pass B (xyzs != NULL) writes xyzs/dirs/ts(/ldirs) for rays[n] = (offset, count).
 * Call with xyzs == NULL for pass A, then again with buffers for pass B, exactly as
 * raymarching.py:301-311 drives the reference. */
void orc_march_rays_train(const float *rays_o, const float *rays_d, const float *rays_ldir,
                          const uint8_t *grid, float bound, int contract, float dt_gamma,
                          uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float *nears,
                          const float *fars, float *xyzs, float *dirs, float *ts, float *ldirs,
                          int32_t *rays, int32_t *counter, const float *noises)
{
    const int first = (xyzs == NULL);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        march_t m;
        march_setup(&m, rays_o + 3 * n, rays_d + 3 * n, 0, bound, contract, dt_gamma, max_steps, C, H, grid);
        uint32_t budget = max_steps;
        float *px = NULL, *pd = NULL, *pt = NULL, *pl = NULL;
        if (!first) {
            const uint32_t off = (uint32_t)rays[2 * n];
            budget = (uint32_t)rays[2 * n + 1];
            px = xyzs + (size_t)off * 3;
            pd = dirs + (size_t)off * 3;
            pt = ts + (size_t)off * 2;
            if (rays_ldir) pl = ldirs + (size_t)off * 3;
        }
        const float far = fars[n];
        float t = nears[n];
        t = fmaf(clampf(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
        uint32_t step = 0;
        while (t < far && step < budget) {
            float p[3], dt;
            if (march_probe(&m, &t, &dt, p)) {
                t += dt;
                if (!first) {
                    px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
                    pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
                    pt[0] = t; pt[1] = dt;
                    px += 3; pd += 3; pt += 2;
                    if (pl) {
                        pl[0] = rays_ldir[3 * n]; pl[1] = rays_ldir[3 * n + 1]; pl[2] = rays_ldir[3 * n + 2];
                        pl += 3;
                    }
                }
                step++;
            }
        }
        if (first) rays[2 * n + 1] = (int32_t)step;
    }
    if (first) {
        uint32_t run = (uint32_t)counter[0];
        for (uint32_t n = 0; n < N; n++) {
            rays[2 * n] = (int32_t)run;
            run += (uint32_t)rays[2 * n + 1];
        }
        counter[0] = (int32_t)run;
    }
}

/* raymarching.cu:519-597 */
void orc_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *ts,
                                      const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                      float *weights, float *weights_sum, float *depth, float *image)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
        if (cnt != 0 && off + cnt <= M) {
            for (uint32_t i = off; i < off + cnt; i++) {
                const float alpha = 1.0f - expf(-sigmas[i] * ts[2 * i + 1]);
                const float w = alpha * T;
                weights[i] = w;
                r = fmaf(w, rgbs[3 * i], r);
                g = fmaf(w, rgbs[3 * i + 1], g);
                b = fmaf(w, rgbs[3 * i + 2], b);
                ws += w;
                d = fmaf(w, ts[2 * i], d);
                T *= 1.0f - alpha;
                if (T < T_thresh) break;
            }
        }
        weights_sum[n] = ws;
        depth[n] = d;
        image[3 * n] = r;
        image[3 * n + 1] = g;
        image[3 * n + 2] = b;
    }
}

/* raymarching.cu:623-712 */
void orc_composite_rays_train_backward(const float *grad_weights, const float *grad_weights_sum,
                                       const float *grad_depth, const float *grad_image,
                                       const float *sigmas, const float *rgbs, const float *ts,
                                       const int32_t *rays, const float *weights_sum, const float *depth,
                                       const float *image, uint32_t M, uint32_t N, float T_thresh,
                                       float *grad_sigmas, float *grad_rgbs)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
        if (cnt == 0 || off + cnt > M) continue;
        const float gr = grad_image[3 * n], gg = grad_image[3 * n + 1], gb = grad_image[3 * n + 2];
        const float gws = grad_weights_sum[n], gd = grad_depth[n];
        const float rF = image[3 * n], gF = image[3 * n + 1], bF = image[3 * n + 2];
        const float wsF = weights_sum[n], dF = depth[n];
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
        for (uint32_t i = off; i < off + cnt; i++) {
            const float dt = ts[2 * i + 1], tm = ts[2 * i];
            const float alpha = 1.0f - expf(-sigmas[i] * dt);
            const float w = alpha * T;
            r = fmaf(w, rgbs[3 * i], r);
            g = fmaf(w, rgbs[3 * i + 1], g);
            b = fmaf(w, rgbs[3 * i + 2], b);
            ws += w;
            d = fmaf(w, tm, d);
            T *= 1.0f - alpha; /* the gradient below uses T AFTER this update */
            grad_rgbs[3 * i] = gr * w;
            grad_rgbs[3 * i + 1] = gg * w;
            grad_rgbs[3 * i + 2] = gb * w;
            float s = gr * (fmaf(T, rgbs[3 * i], -(rF - r)));
            s = fmaf(gg, fmaf(T, rgbs[3 * i + 1], -(gF - g)), s);
            s = fmaf(gb, fmaf(T, rgbs[3 * i + 2], -(bF - b)), s);
            s = fmaf(gws + grad_weights[i], T - (wsF - ws), s);
            s = fmaf(gd, fmaf(T, tm, -(dF - d)), s);
            grad_sigmas[i] = dt * s;
            if (T < T_thresh) break;
        }
    }
}

/* raymarching.cu:731-846 */
void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                    const float *rays_o, const float *rays_d, float bound, int contract, float dt_gamma,
                    uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid, const float *nears,
                    const float *fars, float *xyzs, float *dirs, float *ts, const float *noises)
{
    (void)nears;
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int32_t ray = rays_alive[n];
        march_t m;
        march_setup(&m, rays_o + 3 * (size_t)ray, rays_d + 3 * (size_t)ray, 1, bound, contract, dt_gamma,
                    max_steps, C, H, grid);
        float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3;
        float *pt = ts + (size_t)n * n_step * 2;
        const float far = fars[ray];
        float t = rays_t[ray];
        t = fmaf(clampf(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
        uint32_t step = 0;
        while (t < far && step < n_step) {
            float p[3], dt;
            if (march_probe(&m, &t, &dt, p)) {
                t += dt;
                px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
                pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
                pt[0] = t; pt[1] = dt;
                px += 3; pd += 3; pt += 2;
                step++;
            }
        }
    }
}

/* raymarching.cu:860-941 */
void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                        float *rays_t, const float *sigmas, const float *rgbs, const float *ts,
                        float *weights_sum, float *depth, float *image)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int32_t ray = rays_alive[n];
        const float *sg = sigmas + (size_t)n * n_step, *cl = rgbs + (size_t)n * n_step * 3;
        const float *tt = ts + (size_t)n * n_step * 2;
        float t = 0.0f;
        float d = depth[ray], r = image[3 * ray], g = image[3 * ray + 1], b = image[3 * ray + 2];
        float ws = weights_sum[ray];
        uint32_t step = 0;
        while (step < n_step) {
            if (tt[0] == 0.0f) break;
            const float alpha = 1.0f - expf(-sg[0] * tt[1]);
            const float T = 1.0f - ws;
            const float w = alpha * T;
            ws += w;
            t = tt[0];
            d = fmaf(w, t, d);
            r = fmaf(w, cl[0], r);
            g = fmaf(w, cl[1], g);
            b = fmaf(w, cl[2], b);
            if (T < T_thresh) break;
            sg++;
            cl += 3;
            tt += 2;
            step++;
        }
        if (step < n_step)
            rays_alive[n] = -1;
        else
            rays_t[ray] = t;
        weights_sum[ray] = ws;
        depth[ray] = d;
        image[3 * ray] = r;
        image[3 * ray + 1] = g;
        image[3 * ray + 2] = b;
    }
}

/* raymarching.py:319-329 (torch_scatter.segment_csr over ray-ordered samples):
 *   d rays_o = sum_i dL/dxyz_i ;  d rays_d = sum_i (dL/dxyz_i * t_i + dL/ddir_i) */
void orc_march_rays_train_backward(const float *grad_xyzs, const float *grad_dirs, const float *ts,
                                   const int32_t *rays, uint32_t N, uint32_t M, float *grad_rays_o,
                                   float *grad_rays_d)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
        double so[3] = {0, 0, 0}, sd[3] = {0, 0, 0};
        if (off + cnt <= M)
            for (uint32_t i = off; i < off + cnt; i++)
                for (int k = 0; k < 3; k++) {
                    so[k] += grad_xyzs[3 * i + k];
                    sd[k] += (double)(grad_xyzs[3 * i + k] * ts[2 * i]) + (grad_dirs ? grad_dirs[3 * i + k] : 0.0f);
                }
        for (int k = 0; k < 3; k++) {
            grad_rays_o[3 * n + k] = (float)so[k];
            grad_rays_d[3 * n + k] = (float)sd[k];
        }
    }
}
