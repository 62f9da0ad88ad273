/* oracle/cpu_port.c -- C / OpenMP restatement of the SMC hot path.  TEST INFRASTRUCTURE + CPU BASELINE ONLY.
 *
 * Nothing under python-qinfer_amd/ may load this: it is used by tests/ (as a second checker, pinned against the
 * reference's golden trajectories in tests/golden/ and against oracle/np_oracle.py) and by bench.py's `cpu_baseline`
 * leg, which times it on the GPU box's host cores (kind "port": SURVEY.md 7 step 1 / 8(d) "libqsmc_cpu").
 *
 * What it restates (file:line relative to /root/reference/src/qinfer/):
 *   SMCUpdater.update            smc.py:388-457      likelihood, w *= L, norm = sum, w /= norm, guards, n_ess
 *   hypothetical_update          smc.py:353-373      |norm| < eps -> 1
 *   n_ess                        distributions.py:299-307
 *   _maybe_resample / resample   smc.py:263-277, 491-551
 *   LiuWestResampler.__call__    resamplers.py:256-392  (incl. quirk Q1 `mus = mus[:k]`, :371-372)
 *   particle_meanfn / covariance distributions.py:337-399
 *   sqrtm_psd                    utils.py:593-607    (eigh -> cyclic Jacobi here)
 *   SimplePrecessionModel        test_models.py:123-143
 *   BinomialModel                derived_models.py:314-329 + utils.py:106-111 (closed-form pmf, as np_oracle.binom_pmf)
 *   RandomizedBenchmarkingModel  rb.py:149-195
 *   TomographyModel              tomography/models.py:149-226 (canonicalize: Hermitian eigendecomposition by Jacobi)
 *
 * RNG modes:
 *   0  MT19937 + NumPy's legacy constructions (random_sample, legacy_gauss polar method with its cache), drawn
 *      serially in the reference's order and shapes: stream-identical to `np.random.seed(s)`;
 *   1  Philox4x32-10 counter streams, one per output particle: the same algorithm with the draws made in parallel
 *      (what the all-cores baseline needs -- a serial generator would be the Amdahl bottleneck, not the path);
 *   2  replay: the caller supplies every uniform / normal in consumption order (the recorded draws of the golden
 *      fixtures), which pins this file against the reference's own trajectories.
 *
 * Layout: the caller hands locations as the reference holds them, (N, d) C-order; internally SoA x[d][N].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QCPU_MAX_D 16
#define QCPU_EPS 2.220446049250313e-16

enum { QCPU_PRECESSION = 1, QCPU_BINOMIAL_PRECESSION = 2, QCPU_RB = 3, QCPU_TOMOGRAPHY = 5, QCPU_BINOMIAL_RB = 6 };

typedef struct {
    int32_t kind, d;
    int64_t n;                 /* particles */
    int32_t n_data;
    const double *ep_t;        /* [n_data] precession time t (kinds 1, 2) */
    const uint64_t *ep_m;      /* [n_data] RB sequence length (kinds 3, 6) */
    const uint64_t *ep_nmeas;  /* [n_data] binomial n_meas (kinds 2, 6) */
    const double *ep_meas;     /* [n_data][d] tomography measurement vectors (kind 5) */
    const int64_t *outcomes;   /* [n_data] */
    double a, h, resample_thresh, min_freq, zero_cov_comp;
    int32_t maxiter, postselect, legacy_q1, canonicalize;
    int32_t rng_mode;          /* 0 MT19937 legacy, 1 Philox parallel, 2 replay */
    uint64_t seed;
    const double *replay;      /* mode 2: flat draw log */
    int64_t replay_len;
    int32_t threads;           /* 0 = OpenMP default */
    const double *basis;       /* tomography: (d, dim, dim) complex128 interleaved (re, im) */
    int32_t dim;
    int32_t check_every;       /* resample test every k data (1 = update(); k > 1 = batch_update cadence) */
} qcpu_job_t;

typedef struct {
    double wall_s;             /* update loop only (prior sampling and layout conversion excluded) */
    double update_s, resample_s;
    int32_t resample_count, status, threads_used;
    int64_t n_failed, replay_used;
    double min_n_ess;
    double mean[QCPU_MAX_D];
    double cov[QCPU_MAX_D * QCPU_MAX_D];
} qcpu_result_t;

/* ---------------------------------------------------------------------------------------------------------------
 * RNG
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t mt[624];
    int idx;
    int has_gauss;
    double gauss;
} mt_t;

static void mt_seed(mt_t *s, uint32_t seed) {                      /* init_genrand (Knuth 1812433253) */
    s->mt[0] = seed;
    for (int i = 1; i < 624; ++i) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
    s->has_gauss = 0;
    s->gauss = 0.0;
}

static uint32_t mt_u32(mt_t *s) {
    if (s->idx >= 624) {
        uint32_t *mt = s->mt;
        int k;
        for (k = 0; k < 624 - 397; ++k) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
            mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; k < 623; ++k) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
            mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        const uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static double mt_double(mt_t *s) {                                 /* random_sample: 53 bits from two outputs */
    const uint32_t a = mt_u32(s) >> 5, b = mt_u32(s) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

static double mt_gauss(mt_t *s) {                                  /* legacy_gauss: polar method, cached second value */
    if (s->has_gauss) {
        s->has_gauss = 0;
        const double t = s->gauss;
        s->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * mt_double(s) - 1.0;
        x2 = 2.0 * mt_double(s) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    s->gauss = f * x1;
    s->has_gauss = 1;
    return f * x2;
}

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

static inline double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

typedef struct {
    int mode;
    mt_t mt;
    uint64_t seed, epoch;
    const double *replay;
    int64_t replay_len, replay_pos;
} rng_t;

/* n uniforms for output particles [0, n): the reference's `np.random.random((n,))` */
static int draw_uniform(rng_t *g, double *u, int64_t n) {
    if (g->mode == 0) {
        for (int64_t i = 0; i < n; ++i) u[i] = mt_double(&g->mt);
    } else if (g->mode == 2) {
        if (g->replay_pos + n > g->replay_len) return -1;
        memcpy(u, g->replay + g->replay_pos, (size_t)n * sizeof(double));
        g->replay_pos += n;
    } else {
        const uint32_t k0 = (uint32_t)g->seed, k1 = (uint32_t)(g->seed >> 32);
        const uint32_t ep = (uint32_t)g->epoch;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            uint32_t c[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32), ep, 0u};
            philox4x32_10(c, k0, k1);
            u[i] = u53(c[0], c[1]);
        }
    }
    return 0;
}

/* z[m][r], m < d, r < k (param-major, ld = k): the reference's `np.random.randn(d, k)`; `round` separates redraws,
 * `ids[r]` (or r) names the output particle whose stream a Philox draw comes from */
static int draw_normal(rng_t *g, double *z, int d, int64_t k, int round, const int64_t *ids) {
    if (g->mode == 0) {
        for (int64_t i = 0; i < (int64_t)d * k; ++i) z[i] = mt_gauss(&g->mt);
    } else if (g->mode == 2) {
        if (g->replay_pos + (int64_t)d * k > g->replay_len) return -1;
        memcpy(z, g->replay + g->replay_pos, (size_t)d * (size_t)k * sizeof(double));
        g->replay_pos += (int64_t)d * k;
    } else {
        const uint32_t k0 = (uint32_t)g->seed, k1 = (uint32_t)(g->seed >> 32);
        const uint32_t ep = (uint32_t)g->epoch;
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < k; ++r) {
            const uint64_t id = (uint64_t)(ids ? ids[r] : r);
            for (int m = 0; m < d; m += 2) {
                uint32_t c[4] = {(uint32_t)id, (uint32_t)(id >> 32), ep, ((uint32_t)round << 8) | (uint32_t)(1 + m / 2)};
                philox4x32_10(c, k0, k1);
                const double u0 = u53(c[0], c[1]), u1 = u53(c[2], c[3]);
                const double rad = sqrt(-2.0 * log(1.0 - u0));          /* Box-Muller: 1 - u0 in (0, 1] */
                const double ang = 6.283185307179586 * u1;
                z[(int64_t)m * k + r] = rad * cos(ang);
                if (m + 1 < d) z[(int64_t)(m + 1) * k + r] = rad * sin(ang);
            }
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * models
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int kind, d;
    double t, m, n_meas, comb, min_freq;
    int64_t outcome;
    const double *meas;
} exp_t;

static double exact_comb(uint64_t n, uint64_t k) {                 /* math.comb(n, k) rounded to double */
    if (k > n) return 0.0;
    if (k > n - k) k = n - k;
    long double c = 1.0L;
    for (uint64_t j = 1; j <= k; ++j) {
        c = c * (long double)(n - k + j) / (long double)j;
        if (c > 1.0e305L) return INFINITY;
    }
    return (double)c;
}

static inline double two_outcome(double pr0, int64_t o) { return o == 0 ? pr0 : 1.0 - pr0; }

static inline double binom_pmf(const exp_t *e, double p) {         /* np_oracle.binom_pmf: c * p**k * (1-p)**(n-k) */
    const double k = (double)e->outcome;
    if (e->outcome < 0 || k > e->n_meas) return 0.0;
    if (e->n_meas <= 64.0) return e->comb * pow(p, k) * pow(1.0 - p, e->n_meas - k);
    /* many measurements: the powers underflow (and the coefficient overflows) long before the product: log space */
    const double lc = lgamma(e->n_meas + 1.0) - lgamma(k + 1.0) - lgamma(e->n_meas - k + 1.0);
    const double lp = (k > 0.0 ? k * log(p) : 0.0) + (e->n_meas - k > 0.0 ? (e->n_meas - k) * log1p(-p) : 0.0);
    return exp(lc + lp);
}

static inline double lik_one(const exp_t *e, const double *p) {
    switch (e->kind) {
        case QCPU_PRECESSION: {
            const double c = cos(e->t * p[0] / 2.0);                   /* test_models.py:134-141 (w_ = 0) */
            return two_outcome(c * c, e->outcome);
        }
        case QCPU_BINOMIAL_PRECESSION: {
            const double c = cos(e->t * p[0] / 2.0);
            return binom_pmf(e, 1.0 - c * c);                          /* pr1 = L(outcome 1) = 1 - pr0 */
        }
        case QCPU_RB: {
            const double pr0 = 1.0 - (p[1] * pow(p[0], e->m) + p[2]);  /* rb.py:190-193 */
            return two_outcome(pr0, e->outcome);
        }
        case QCPU_BINOMIAL_RB: {
            const double pr0 = 1.0 - (p[1] * pow(p[0], e->m) + p[2]);
            return binom_pmf(e, 1.0 - pr0);
        }
        case QCPU_TOMOGRAPHY: {
            double s = 0.0;
            for (int i = 0; i < e->d; ++i) s += e->meas[i] * p[i];     /* tomography/models.py:216-226 */
            const double pr1 = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
            return two_outcome(1.0 - pr1, e->outcome);
        }
    }
    return 0.0;
}

static inline int valid_one(int kind, double min_freq, const double *p) {
    switch (kind) {
        case QCPU_PRECESSION:
        case QCPU_BINOMIAL_PRECESSION: return p[0] > min_freq;          /* test_models.py:109-110 */
        case QCPU_RB:
        case QCPU_BINOMIAL_RB: {                                        /* rb.py:165-176 */
            const double P = p[0], A = p[1], B = p[2];
            return 0.0 <= P && P <= 1.0 && 0.0 <= A && A <= 1.0 && 0.0 <= B && B <= 1.0 && A + B <= 1.0 && A * P + B <= 1.0;
        }
        default: return 1;                                              /* tomography/models.py:143-147 */
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * small dense linear algebra (host, d <= 16)
 * ------------------------------------------------------------------------------------------------------------- */
/* S = scale * sqrtm_psd(A) (utils.py:593-607): symmetric cyclic Jacobi, eigenvalues <= 0 clamped; returns ||S S - A||_F */
static double sqrtm_psd(const double *A, int n, double scale, double *S) {
    double a[QCPU_MAX_D * QCPU_MAX_D], v[QCPU_MAX_D * QCPU_MAX_D], sq[QCPU_MAX_D * QCPU_MAX_D];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            a[i * n + j] = 0.5 * (A[i * n + j] + A[j * n + i]);
            v[i * n + j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += a[i * n + i] * a[i * n + i];
            for (int j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
        }
        if (off == 0.0 || off <= 1e-34 * diag) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[p * n + q];
                if (apq == 0.0) continue;
                const double tau = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq;
                    a[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk;
                    a[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq;
                    v[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) {
                const double lam = a[k * n + k];
                s += v[i * n + k] * (lam <= 0.0 ? 0.0 : sqrt(lam)) * v[j * n + k];
            }
            sq[i * n + j] = s;
        }
    double e2 = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += sq[i * n + k] * sq[k * n + j];
            e2 += (s - A[i * n + j]) * (s - A[i * n + j]);
        }
    for (int k = 0; k < n * n; ++k) S[k] = scale * sq[k];
    return sqrt(e2);
}

/* tomography canonicalize of one particle (tomography/models.py:149-209), dim x dim complex Hermitian Jacobi */
#define TDIM_MAX 4
static void tomo_canon_one(const double *basis, int dim, int allow_subnormalized, double *p) {
    const int D = dim * dim;
    double Ar[TDIM_MAX][TDIM_MAX], Ai[TDIM_MAX][TDIM_MAX], Vr[TDIM_MAX][TDIM_MAX], Vi[TDIM_MAX][TDIM_MAX];
    for (int r = 0; r < dim; ++r)
        for (int c = 0; c < dim; ++c) {
            double sr = 0.0, si = 0.0;
            for (int a = 0; a < D; ++a) {
                sr += p[a] * basis[2 * ((a * dim + r) * dim + c)];
                si += p[a] * basis[2 * ((a * dim + r) * dim + c) + 1];
            }
            Ar[r][c] = sr; Ai[r][c] = si;
            Vr[r][c] = r == c ? 1.0 : 0.0; Vi[r][c] = 0.0;
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag2 = 0.0;
        for (int r = 0; r < dim; ++r) {
            diag2 += Ar[r][r] * Ar[r][r];
            for (int c = r + 1; c < dim; ++c) off += Ar[r][c] * Ar[r][c] + Ai[r][c] * Ai[r][c];
        }
        if (off <= 1e-34 * diag2) break;
        for (int pI = 0; pI < dim; ++pI)
            for (int q = pI + 1; q < dim; ++q) {
                const double hr = Ar[pI][q], hi = Ai[pI][q];
                const double mag = sqrt(hr * hr + hi * hi);
                if (mag < 1e-300) continue;
                const double er = hr / mag, ei = hi / mag;
                const double tau = (Ar[q][q] - Ar[pI][pI]) / (2.0 * mag);
                const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = tt * cs;
                for (int r = 0; r < dim; ++r) {
                    const double apr = Ar[r][pI], api = Ai[r][pI], aqr = Ar[r][q], aqi = Ai[r][q];
                    Ar[r][pI] = cs * apr - sn * (er * aqr + ei * aqi);
                    Ai[r][pI] = cs * api - sn * (er * aqi - ei * aqr);
                    Ar[r][q] = sn * (er * apr - ei * api) + cs * aqr;
                    Ai[r][q] = sn * (er * api + ei * apr) + cs * aqi;
                    const double vpr = Vr[r][pI], vpi = Vi[r][pI], vqr = Vr[r][q], vqi = Vi[r][q];
                    Vr[r][pI] = cs * vpr - sn * (er * vqr + ei * vqi);
                    Vi[r][pI] = cs * vpi - sn * (er * vqi - ei * vqr);
                    Vr[r][q] = sn * (er * vpr - ei * vpi) + cs * vqr;
                    Vi[r][q] = sn * (er * vpi + ei * vpr) + cs * vqi;
                }
                for (int c2 = 0; c2 < dim; ++c2) {
                    const double apr = Ar[pI][c2], api = Ai[pI][c2], aqr = Ar[q][c2], aqi = Ai[q][c2];
                    Ar[pI][c2] = cs * apr - sn * (er * aqr - ei * aqi);
                    Ai[pI][c2] = cs * api - sn * (er * aqi + ei * aqr);
                    Ar[q][c2] = sn * (er * apr + ei * api) + cs * aqr;
                    Ai[q][c2] = sn * (er * api - ei * apr) + cs * aqi;
                }
            }
    }
    int any_neg = 0;
    double lam[TDIM_MAX];
    for (int r = 0; r < dim; ++r) {
        lam[r] = Ar[r][r];
        any_neg |= !(lam[r] >= 0.0);
    }
    if (any_neg) {
        double Rr[TDIM_MAX][TDIM_MAX], Ri[TDIM_MAX][TDIM_MAX];
        for (int r = 0; r < dim; ++r) lam[r] = lam[r] < 0.0 ? 0.0 : lam[r];
        for (int r = 0; r < dim; ++r)
            for (int c = 0; c < dim; ++c) {
                double sr = 0.0, si = 0.0;
                for (int k = 0; k < dim; ++k) {
                    sr += lam[k] * (Vr[r][k] * Vr[c][k] + Vi[r][k] * Vi[c][k]);
                    si += lam[k] * (Vi[r][k] * Vr[c][k] - Vr[r][k] * Vi[c][k]);
                }
                Rr[r][c] = sr; Ri[r][c] = si;
            }
        for (int a = 0; a < D; ++a) {
            double s = 0.0;
            for (int r = 0; r < dim; ++r)
                for (int c = 0; c < dim; ++c)
                    s += basis[2 * ((a * dim + r) * dim + c)] * Rr[r][c] + basis[2 * ((a * dim + r) * dim + c) + 1] * Ri[r][c];
            p[a] = s;
        }
    }
    if (!allow_subnormalized) {
        const double nrm = p[0] * sqrt((double)dim);
        for (int a = 0; a < D; ++a) p[a] = p[a] / nrm;
    }
}

static void canonicalize_cloud(const qcpu_job_t *job, double *x, int64_t n) {
    if (job->kind != QCPU_TOMOGRAPHY || !job->canonicalize || !job->basis) return;
    const int d = job->d;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double p[QCPU_MAX_D];
        for (int m = 0; m < d; ++m) p[m] = x[(int64_t)m * n + i];
        tomo_canon_one(job->basis, job->dim, 0, p);
        for (int m = 0; m < d; ++m) x[(int64_t)m * n + i] = p[m];
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * the SMC loop
 * ------------------------------------------------------------------------------------------------------------- */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* inclusive scan of w into cdf: sequential (np.cumsum order) with one thread, blocked two-pass otherwise */
static void cumsum(const double *w, double *cdf, int64_t n, int threads) {
    if (threads <= 1) {
        double run = 0.0;
        for (int64_t i = 0; i < n; ++i) { run += w[i]; cdf[i] = run; }
        return;
    }
    const int T = threads;
    double *tot = (double *)calloc((size_t)T + 1, sizeof(double));
#pragma omp parallel num_threads(T)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        double run = 0.0;
        for (int64_t i = lo; i < hi; ++i) { run += w[i]; cdf[i] = run; }
        tot[t + 1] = run;
#pragma omp barrier
#pragma omp single
        for (int k = 1; k <= T; ++k) tot[k] += tot[k - 1];
        const double off = tot[t];
        if (t > 0)
            for (int64_t i = lo; i < hi; ++i) cdf[i] += off;
    }
    free(tot);
}

static int resample(const qcpu_job_t *job, rng_t *g, double **px, double *w, int64_t n, int threads,
                    double *scratch_cdf, double *scratch_u, int64_t *js, double *mus, double *z, int64_t *idxs,
                    double *xnew, int64_t *n_failed) {
    const int d = job->d;
    double *x = *px;
    /* mean, covariance (distributions.py:337-399): E[x x^T] - mu mu^T */
    double mu[QCPU_MAX_D], second[QCPU_MAX_D * QCPU_MAX_D], cov[QCPU_MAX_D * QCPU_MAX_D], S[QCPU_MAX_D * QCPU_MAX_D];
    for (int m = 0; m < d; ++m) {
        const double *xm = x + (int64_t)m * n;
        double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
        for (int64_t i = 0; i < n; ++i) s += w[i] * xm[i];
        mu[m] = s;
        for (int q = m; q < d; ++q) {
            const double *xq = x + (int64_t)q * n;
            double s2 = 0.0;
#pragma omp parallel for reduction(+ : s2) schedule(static)
            for (int64_t i = 0; i < n; ++i) s2 += w[i] * xm[i] * xq[i];
            second[m * d + q] = second[q * d + m] = s2;
        }
    }
    double fro = 0.0;
    for (int m = 0; m < d; ++m)
        for (int q = 0; q < d; ++q) {
            cov[m * d + q] = second[m * d + q] - mu[m] * mu[q];
            fro += cov[m * d + q] * cov[m * d + q];
        }
    if (fro == 0.0)                                                /* resamplers.py:283-294 */
        for (int m = 0; m < d; ++m)
            for (int q = 0; q < d; ++q) cov[m * d + q] = m == q ? job->zero_cov_comp : 0.0;
    const double err = sqrtm_psd(cov, d, job->h, S);
    if (!isfinite(err)) return -3;
    cumsum(w, scratch_cdf, n, threads);                            /* :308 */
    const int64_t n_out = n;
    g->epoch += 1;
    if (draw_uniform(g, scratch_u, n_out)) return -4;
    const double a = job->a;
    /* js = cdf.searchsorted(u, side='right') (:318-321; the reference does not clamp -- Q2 -- an index n would be
     * out of bounds there, so clamp like the device code); mus = a x[js] + (1 - a) mu (:325) */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_out; ++i) {
        const double u = scratch_u[i];
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (scratch_cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int64_t j = lo < n - 1 ? lo : n - 1;
        js[i] = j;
        for (int m = 0; m < d; ++m) mus[(int64_t)m * n_out + i] = a * x[(int64_t)m * n + j] + (1.0 - a) * mu[m];
    }
    int64_t k = n_out;
    int first = 1;
    int rounds = 0;
    while (k > 0 && rounds < job->maxiter) {                       /* :327-372 */
        ++rounds;
        if (draw_normal(g, z, d, k, rounds - 1, first ? NULL : idxs)) return -4;
        /* new[idxs[r]] = mus[c] + (S z)[:, r]; Q1: after the first round the reference keeps the FIRST k centres */
        int64_t n_bad = 0;
        const int legacy = job->legacy_q1;
        /* (two passes: compute + validity flag in parallel, then a serial stable compaction like np.nonzero) */
        unsigned char *ok = (unsigned char *)scratch_u;            /* reuse: u is dead after the search */
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < k; ++r) {
            const int64_t dst = first ? r : idxs[r];
            const int64_t c = (first || !legacy) ? dst : r;
            double p[QCPU_MAX_D];
            for (int m = 0; m < d; ++m) {
                double s = 0.0;
                for (int q = 0; q < d; ++q) s += S[m * d + q] * z[(int64_t)q * k + r];
                p[m] = mus[(int64_t)m * n_out + c] + s;
                xnew[(int64_t)m * n_out + dst] = p[m];
            }
            ok[r] = (unsigned char)(!job->postselect || valid_one(job->kind, job->min_freq, p));
        }
        for (int64_t r = 0; r < k; ++r)
            if (!ok[r]) idxs[n_bad++] = first ? r : idxs[r];
        first = 0;
        k = n_bad;
    }
    *n_failed += k;
    /* swap clouds; uniform weights (:390) */
    memcpy(x, xnew, (size_t)d * (size_t)n_out * sizeof(double));
    const double w0 = 1.0 / (double)n_out;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_out; ++i) w[i] = w0;
    canonicalize_cloud(job, x, n_out);                             /* smc.py:529 */
    return 0;
}

int qcpu_smc_run(const qcpu_job_t *job, double *locs /* (n, d) in/out */, double *weights_out /* [n] */,
                 double *norm_record /* [n_data] */, double *ess_record /* [n_data] */,
                 double *mean_record /* [n_data][d] or NULL */, qcpu_result_t *res) {
    if (!job || !locs || !res || job->d < 1 || job->d > QCPU_MAX_D || job->n < 1) return -1;
    memset(res, 0, sizeof(*res));
    const int d = job->d;
    const int64_t n = job->n;
    int threads = job->threads;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    omp_set_num_threads(threads);
#else
    threads = 1;
#endif
    res->threads_used = threads;
    double *x = (double *)malloc(sizeof(double) * (size_t)d * (size_t)n);
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    double *cdf = (double *)malloc(sizeof(double) * (size_t)n);
    double *u = (double *)malloc(sizeof(double) * (size_t)n);
    int64_t *js = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t *idxs = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    double *mus = (double *)malloc(sizeof(double) * (size_t)d * (size_t)n);
    double *z = (double *)malloc(sizeof(double) * (size_t)d * (size_t)n);
    double *xnew = (double *)malloc(sizeof(double) * (size_t)d * (size_t)n);
    if (!x || !w || !cdf || !u || !js || !idxs || !mus || !z || !xnew) return -5;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        for (int m = 0; m < d; ++m) x[(int64_t)m * n + i] = locs[i * d + m];
        w[i] = 1.0 / (double)n;                                    /* smc.py:307 */
    }
    canonicalize_cloud(job, x, n);                                 /* reset(): smc.py:317-320 */
    rng_t g;
    memset(&g, 0, sizeof(g));
    g.mode = job->rng_mode;
    g.seed = job->seed;
    g.replay = job->replay;
    g.replay_len = job->replay_len;
    if (g.mode == 0) mt_seed(&g.mt, (uint32_t)job->seed);
    double min_ess = (double)n;
    int status = 0;
    const int check_every = job->check_every > 0 ? job->check_every : 1;
    const double t_begin = now_s();
    double t_upd = 0.0, t_rs = 0.0;
    for (int k = 0; k < job->n_data && status == 0; ++k) {
        const double t0 = now_s();
        exp_t e;
        memset(&e, 0, sizeof(e));
        e.kind = job->kind;
        e.d = d;
        e.outcome = job->outcomes[k];
        e.t = job->ep_t ? job->ep_t[k] : 0.0;
        e.m = job->ep_m ? (double)job->ep_m[k] : 0.0;
        e.n_meas = job->ep_nmeas ? (double)job->ep_nmeas[k] : 0.0;
        e.meas = job->ep_meas ? job->ep_meas + (size_t)k * d : NULL;
        if (job->ep_nmeas && e.outcome >= 0) e.comb = exact_comb(job->ep_nmeas[k], (uint64_t)e.outcome);
        /* weights = w * L; norm = sum (smc.py:353-357) */
        double norm = 0.0;
        int64_t n_bad = 0;
#pragma omp parallel for reduction(+ : norm, n_bad) schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            double p[QCPU_MAX_D];
            for (int m = 0; m < d; ++m) p[m] = x[(int64_t)m * n + i];
            const double v = w[i] * lik_one(&e, p);
            w[i] = v;
            norm += v;
            n_bad += !(v >= 0.0);
        }
        const double fixed = fabs(norm) < QCPU_EPS ? 1.0 : norm;   /* :369-370 */
        double ss = 0.0, sum_w = 0.0;
        if (n_bad) {                                               /* :416-418: clip to [0, 1] after normalising */
#pragma omp parallel for reduction(+ : ss, sum_w) schedule(static)
            for (int64_t i = 0; i < n; ++i) {
                double v = w[i] / fixed;
                if (v == v) v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
                w[i] = v;
                ss += v * v;
                sum_w += v;
            }
        } else {
#pragma omp parallel for reduction(+ : ss, sum_w) schedule(static)
            for (int64_t i = 0; i < n; ++i) {
                const double v = w[i] / fixed;
                w[i] = v;
                ss += v * v;
                sum_w += v;
            }
        }
        if (!(sum_w > 10.0 * QCPU_EPS)) { status = -2; break; }   /* zero_weight_policy 'error' (:423-436) */
        norm_record[k] = norm;
        const double ess = 1.0 / ss;                               /* distributions.py:299-307 */
        ess_record[k] = ess;
        if (ess <= min_ess) min_ess = ess;
        const double t1 = now_s();
        t_upd += t1 - t0;
        if ((k + 1) % check_every == 0 && ess < (double)n * job->resample_thresh) {   /* smc.py:263-277 */
            status = resample(job, &g, &x, w, n, threads, cdf, u, js, mus, z, idxs, xnew, &res->n_failed);
            res->resample_count += 1;
            t_rs += now_s() - t1;
        }
        if (mean_record)
            for (int m = 0; m < d; ++m) {
                const double *xm = x + (int64_t)m * n;
                double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
                for (int64_t i = 0; i < n; ++i) s += w[i] * xm[i];
                mean_record[(size_t)k * d + m] = s;
            }
    }
    res->wall_s = now_s() - t_begin;
    res->update_s = t_upd;
    res->resample_s = t_rs;
    res->status = status;
    res->min_n_ess = min_ess;
    res->replay_used = g.replay_pos;
    /* final read-outs */
    for (int m = 0; m < d; ++m) {
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += w[i] * x[(int64_t)m * n + i];
        res->mean[m] = s;
    }
    for (int m = 0; m < d; ++m)
        for (int q = 0; q < d; ++q) {
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += w[i] * x[(int64_t)m * n + i] * x[(int64_t)q * n + i];
            res->cov[m * d + q] = s - res->mean[m] * res->mean[q];
        }
    for (int64_t i = 0; i < n; ++i) {
        for (int m = 0; m < d; ++m) locs[i * d + m] = x[(int64_t)m * n + i];
        if (weights_out) weights_out[i] = w[i];
    }
    free(x); free(w); free(cdf); free(u); free(js); free(idxs); free(mus); free(z); free(xnew);
    return status;
}

/* known-answer hooks for the tests: the first n outputs of the legacy stream after np.random.seed(seed) */
void qcpu_mt_random(uint32_t seed, double *out, int64_t n) {
    mt_t s;
    mt_seed(&s, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = mt_double(&s);
}

void qcpu_mt_randn(uint32_t seed, double *out, int64_t n) {
    mt_t s;
    mt_seed(&s, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = mt_gauss(&s);
}

/* contract likelihood L[i] for one experiment / outcome (pins lik_one against the G2 golden vectors) */
int qcpu_likelihood(int kind, int d, const double *locs /* (n, d) */, int64_t n, double t, uint64_t m, uint64_t n_meas,
                    const double *meas, int64_t outcome, double *L) {
    exp_t e;
    memset(&e, 0, sizeof(e));
    e.kind = kind; e.d = d; e.t = t; e.m = (double)m; e.n_meas = (double)n_meas; e.meas = meas; e.outcome = outcome;
    if (n_meas && outcome >= 0) e.comb = exact_comb(n_meas, (uint64_t)outcome);
    for (int64_t i = 0; i < n; ++i) L[i] = lik_one(&e, locs + i * d);
    return 0;
}

int qcpu_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
