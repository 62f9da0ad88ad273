// qsmc_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Everything here is written for CDNA4: 64-lane wavefronts (hard-coded, never warpSize-agnostic),
// 256-thread workgroups (4 waves, one per SIMD), float64 arithmetic with contraction OFF so that
// products and sums round exactly like the NumPy expressions they replace.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qsmc.h"

#define QSMC_BLOCK 256
#define QSMC_WAVE 64
#define QSMC_WAVES_PER_BLOCK (QSMC_BLOCK / QSMC_WAVE)
#define QSMC_GRID_CAP 2048            // 256 CUs x 8 resident 256-thread blocks

namespace qsmc {

// ---------------------------------------------------------------------------------------------
// Experiment parameters as the kernels see them (by value in the kernarg segment -> SGPRs).
// ---------------------------------------------------------------------------------------------
struct ExpArgs {
    double t, w_;            // precession
    double n_meas;           // binomial: n as double
    double comb;             // C(n_meas, k) rounded to double (inf -> use log_comb)
    double log_comb;
    double m;                // RB sequence length as double (NumPy: float64 ** uint64 -> pow(double))
    int32_t reference;       // RB interleaved
    int32_t d;
    double lik_pow;          // MLEModel: likelihood ** lik_pow (0 = plain)
    double meas[QSMC_MAX_D]; // tomography
    int32_t nnz;             // tomography: how many entries of meas[0 .. d) are not zero, and which (ascending)
    int32_t nz_idx[QSMC_MAX_D];
};

// ---------------------------------------------------------------------------------------------
// Likelihood of ONE outcome for one particle.  p[] holds the particle's d parameters.
// Each follows the reference expression operation by operation (see qsmc.h for file:line).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ double two_outcome(double pr0, int64_t outcome) {
    // abstract_model.py:666-686: outcome 0 -> pr0, else 1 - pr0
    return outcome == 0 ? pr0 : 1.0 - pr0;
}

// cos(x)^2 for the likelihoods.  The fused update kernel turned out to be VALU-bound, not HBM-bound (its
// 240 MB working set lives in the 256 MB Infinity Cache: a bare read-read-write kernel over it takes 30 us,
// the update took 44), and OCML's full-range fp64 cos was most of its ~140 instructions per particle.
// Only the parity of k = rint(x * 2/pi) matters for the square: cos^2 x = cos^2 r (k even) or sin^2 r
// (k odd), r = x - k pi/2.  Two fused steps of Cody-Waite give r: fma(-k, PIO2_HI, x) is EXACT while
// |x| < 2^33 pi/2 (x and k PIO2_HI are both multiples of 2^-52 there and the difference is < 2), then
// fma(-k, PIO2_LO, r) rounds once; the neglected tail is k * 2^-107 < 2^-74.  fdlibm's __kernel_sin /
// __kernel_cos minimax polynomials on |r| <= pi/4 (+ 1e-6: k may be off by one at a boundary).
// |error| <= ~3 ulp(1) on the square -- inside the stated 4-ulp / 1e-15 likelihood tolerance, checked
// against the reference's own numbers up to t = (9/8)^199 (G2).  Beyond 1e10 rad: the library cos.
// (out of line on purpose: inlined, the library cos and its Payne-Hanek tables cost the update kernel
// 14 VGPRs and a wave of occupancy for a path no lane takes below 1e10 rad)
// ---------------------------------------------------------------------------------------------
// ln and exp for the likelihoods that need them (binomial pmf, RB survival p^m, T2 decay, MLE power).  The
// library's fp64 log / log1p / exp are written for every corner of double and, inlined, made the binomial
// update kernel 7000 VALU instructions (5x the precession one, VALU-bound at twice its time).  The arguments
// here are plain -- probabilities in [0, 1], exponents within +-700 -- so the classic fdlibm reductions do
// (e_log.c, e_exp.c: < 1 ulp), with the coefficients in constant memory (see cos_sq) and everything unusual
// (zero, subnormal, negative, inf, NaN, overflow) sent to the library out of line.  Host code keeps libm.
// ---------------------------------------------------------------------------------------------
#ifdef __HIP_DEVICE_COMPILE__
__device__ __attribute__((noinline)) double log_full_range(double x) { return log(x); }
__device__ __attribute__((noinline)) double exp_full_range(double x) { return exp(x); }
__constant__ double FLOG_K[7] = {1.531383769920937332e-01, 2.222219843214978396e-01, 3.999999999940941908e-01,
                                 1.479819860511658591e-01, 1.818357216161805012e-01, 2.857142874366239149e-01,
                                 6.666666666666735130e-01};
__constant__ double FEXP_K[5] = {4.13813679705723846039e-08, -1.65339022054652515390e-06, 6.61375632143793436117e-05,
                                 -2.77777777770155933842e-03, 1.66666666666666019037e-01};
__device__ __forceinline__ double fast_div(double n, double d) {       // finite d > 0: rcp seed, two Newton steps, one correction
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}
__device__ __forceinline__ double fast_log(double x) {
    if (!(x >= 2.2250738585072014e-308 && x < 1.7976931348623157e308)) return log_full_range(x);
    int hx = __double2hiint(x);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;                   // mantissa >= sqrt(2): use m / 2
    const double m = __hiloint2double(hx | (i ^ 0x3ff00000), __double2loint(x));
    k += i >> 20;
    const double f = m - 1.0;
    const double s = fast_div(f, 2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, FLOG_K[0], FLOG_K[1]), FLOG_K[2]);
    const double t2 = z * fma(w, fma(w, fma(w, FLOG_K[3], FLOG_K[4]), FLOG_K[5]), FLOG_K[6]);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}
// ln(1 - p) for p in [0, 1): w = fl(1 - p) misses 1 - p by exactly d = (w - 1) + p, and ln(w - d) = ln w - d / w to
// first order in d / w <= 2^-53 -- what log1p(-p) gives, to the last bit or two, for the price of one division.
__device__ __forceinline__ double fast_log1m(double p) {
    const double w = 1.0 - p;
    if (!(w > 0.0)) return log_full_range(w);                    // p >= 1 (or NaN)
    const double d = (w - 1.0) + p;
    return fast_log(w) - fast_div(d, w);
}
__device__ __forceinline__ double fast_exp(double x) {
    if (!(x > -708.0 && x < 709.0)) return exp_full_range(x);    // underflow / overflow range, inf, NaN
    const double k = rint(x * 1.44269504088896338700e+00);
    const double hi = fma(-k, 6.93147180369123816490e-01, x);   // k ln2_hi is exact: ln2_hi ends in 32 zero bits
    const double lo = k * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * fma(t, fma(t, fma(t, fma(t, FEXP_K[0], FEXP_K[1]), FEXP_K[2]), FEXP_K[3]), FEXP_K[4]);
    const double y = 1.0 - ((lo - fast_div(r * c, 2.0 - c)) - hi);
    return ldexp(y, (int)k);
}
#else
// (host compilation pass: what host code calls, and what kernels are type-checked against)
__host__ __device__ inline double fast_log(double x) { return log(x); }
__host__ __device__ inline double fast_log1m(double p) { return log1p(-p); }
__host__ __device__ inline double fast_exp(double x) { return exp(x); }
#endif

__host__ __device__ __attribute__((noinline)) double cos_sq_full_range(double x) {
    const double c = cos(x);
    return c * c;
}

// Polynomial coefficients of cos_sq: on the device they come from constant memory, i.e. scalar loads into SGPRs.
// As literals every Horner step compiled to a v_mov_b64 (the constant into the accumulator) + v_fmac_f64 -- 226
// moves in k_update_fused, 15 % of its VALU instructions; as SGPR operands of v_fma_f64 they cost nothing (VALU
// instructions -12 %, 115 -> 95 VGPRs).  The weighted kernel is HBM-bound and does not notice (40.6 us); the
// implicit-weight variant, which streams a third less, went 33.0 -> 31.3 us.  Same values, same results.
#ifdef __HIP_DEVICE_COMPILE__
__constant__ double CSQ_K[12] = {
    1.58969099521155010221e-10,  -2.50507602534068634195e-08, 2.75573137070700676789e-06,      // sin: z^5 .. z^0
    -1.98412698298579493134e-04, 8.33333333332248946124e-03,  -1.66666666666666324348e-01,
    -1.13596475577881948265e-11, 2.08757232129817482790e-09,  -2.75573143513906633035e-07,     // cos: z^5 .. z^0
    2.48015872894767294178e-05,  -1.38888888888741095749e-03, 4.16666666666666019037e-02};
#define CSQ(i, lit) CSQ_K[i]
#else
#define CSQ(i, lit) (lit)
#endif
__host__ __device__ __forceinline__ double cos_sq(double x) {
    const double ax = fabs(x);
    if (!(ax <= 1.0e10)) return cos_sq_full_range(x);      // huge, inf or NaN
    const double k = rint(ax * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632679489655800e+00, ax);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = fma(z, CSQ(0, 1.58969099521155010221e-10), CSQ(1, -2.50507602534068634195e-08));
    ps = fma(z, ps, CSQ(2, 2.75573137070700676789e-06));
    ps = fma(z, ps, CSQ(3, -1.98412698298579493134e-04));
    ps = fma(z, ps, CSQ(4, 8.33333333332248946124e-03));
    ps = fma(z, ps, CSQ(5, -1.66666666666666324348e-01));
    const double sn = fma(r * z, ps, r);
    // |r| <= pi / 4: sin^2 r <= 1 / 2, so 1 - sin^2 r carries no cancellation (absolute error <= 4e-16, like the square of a
    // cosine polynomial -- which this replaced in round 4: 9 fp64 instructions of 30 in kernels that are VALU-bound on the
    // likelihood: k_update_multi, the design kernels)
    const double s2 = sn * sn;
    const double kh = 0.5 * k;
    return (kh != floor(kh)) ? s2 : 1.0 - s2;              // k odd: cos^2 x = sin^2 r
}

// cos_sq for an argument the caller has already bounded by 1e10 (k_update_multi hoists that test out of its K likelihoods
// per particle): the same operations on the same values, so the same bits as cos_sq there.
__host__ __device__ __forceinline__ double cos_sq_inrange(double x) {
    const double ax = fabs(x);
    const double k = rint(ax * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632679489655800e+00, ax);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = fma(z, CSQ(0, 1.58969099521155010221e-10), CSQ(1, -2.50507602534068634195e-08));
    ps = fma(z, ps, CSQ(2, 2.75573137070700676789e-06));
    ps = fma(z, ps, CSQ(3, -1.98412698298579493134e-04));
    ps = fma(z, ps, CSQ(4, 8.33333333332248946124e-03));
    ps = fma(z, ps, CSQ(5, -1.66666666666666324348e-01));
    const double sn = fma(r * z, ps, r);
    const double s2 = sn * sn;
    const double kh = 0.5 * k;
    return (kh != floor(kh)) ? s2 : 1.0 - s2;
}

__host__ __device__ __forceinline__ double precession_pr0(double omega, const ExpArgs &e) {
    // test_models.py:134-141: cos(t * dw / 2) ** 2
    const double dw = omega - e.w_;
    return cos_sq(e.t * dw / 2.0);
}

template <int KIND> struct Model;

template <> struct Model<QSMC_MODEL_PRECESSION> {
    static constexpr int D = 1;
    static __host__ __device__ __forceinline__ double lik(const double *p, const ExpArgs &e, int64_t o) {
        return two_outcome(precession_pr0(p[0], e), o);
    }
    static __host__ __device__ __forceinline__ bool valid(const double *p, double min_freq) {
        return p[0] > min_freq;
    }
};

// x^k for a non-negative integer k that is THE SAME FOR EVERY LANE (an outcome count or n_meas - outcome: kernel
// arguments, i.e. SGPRs): square-and-multiply under scalar control flow, at most 2 log2(k) fp64 multiplies.  Each
// multiply rounds once and a squaring doubles the accumulated relative error, so the result is within ~k/2 ulp.
__host__ __device__ __forceinline__ double powi_uniform(double x, unsigned k) {
    double r = 1.0, b = x;
    while (k) {
        if (k & 1u) r *= b;
        k >>= 1;
        if (k) b *= b;
    }
    return r;
}

// The same for N values at once (one bit loop, N independent multiply chains inside it): per value the very multiplications
// of powi_uniform in the same order -- the same bits -- but the N chains overlap instead of running one after the other
// (a lane of the update kernel holds 8 particles of a tile; round 4).
template <int N>
__host__ __device__ __forceinline__ void powi_uniform_n(const double (&x)[N], unsigned k, double (&r)[N]) {
    double b[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { r[i] = 1.0; b[i] = x[i]; }
    while (k) {
        if (k & 1u) {
#pragma unroll
            for (int i = 0; i < N; ++i) r[i] *= b[i];
        }
        k >>= 1;
        if (k) {
#pragma unroll
            for (int i = 0; i < N; ++i) b[i] *= b[i];
        }
    }
}

// derived_models.py:317-325 + utils.py:106-111: Binom(n_meas, pr1).pmf(k), pr1 = L_underlying(outcome 1)
__host__ __device__ __forceinline__ double binom_pmf(double pr1, const ExpArgs &e, int64_t o) {
    const double k = (double)o;
    if (o < 0 || k > e.n_meas) return 0.0;
    if (e.n_meas <= 64.0) {
        // C(n,k) p^k (1-p)^(n-k) literally, with the two integer powers by square-and-multiply: k and n - k come from
        // the kernel arguments, so the loops are scalar control flow and cost <= ~20 multiplies together -- the
        // log / log1p / exp form below is ~100 fp64 operations and made this the one update kernel that was
        // VALU-bound at twice the precession kernel's time (75 vs 40 us at N = 1e7).  Error <= ~n/2 ulp = 4e-15 at
        // n_meas = 25 (the closed form is held to 1e-12 against SciPy's pmf: G2, G8); the same graceful underflow.
        if (!(pr1 >= 0.0 && pr1 <= 1.0)) return NAN;               // (an invalid particle: SciPy's pmf gives nan too)
        return (e.comb * powi_uniform(pr1, (unsigned)o)) * powi_uniform(1.0 - pr1, (unsigned)(e.n_meas - k));
    }
    // many measurements: exp(ln C + k ln p + (n-k) ln(1-p)), everything in log space -- the powers underflow and the
    // coefficient overflows long before their product does (C(1000, 550) p^550 (1-p)^450 at p = 0.02 is 1e-85 while
    // p^550 alone is 1e-934).  Two logarithms and one exponential; relative error ~ n eps |ln p| + the host's lgamma
    // (~1e-13 at n = 1000, 1e-11 at n = 5000: inside 1e-9 of SciPy's pmf, fixture g2_edges).  0 * ln 0 never forms:
    // a zero exponent drops its term.
    const double lp = (k > 0.0 ? k * fast_log(pr1) : 0.0) + (e.n_meas - k > 0.0 ? (e.n_meas - k) * fast_log1m(pr1) : 0.0);
    return fast_exp(e.log_comb + lp);
}

template <> struct Model<QSMC_MODEL_BINOMIAL_PRECESSION> {
    static constexpr int D = 1;
    static __host__ __device__ __forceinline__ double lik(const double *p, const ExpArgs &e, int64_t o) {
        return binom_pmf(1.0 - precession_pr0(p[0], e), e, o);
    }
    static __host__ __device__ __forceinline__ bool valid(const double *p, double min_freq) {
        return p[0] > min_freq;
    }
};

// p ** m for the RB survival law (rb.py:190: float64 ** uint64), p in [0, 1], m a non-negative integer held in a
// double: exp(m ln p).  The error of this form is ~(|m ln p| + 1) eps of the RESULT, and the result is below 1e-15
// of anything visible once |m ln p| > 35 -- so it stays inside the 4-ulp likelihood tolerance wherever p ** m
// matters (G2 checks m up to 1e5), at about two thirds of the instructions of a full fp64 pow().
__host__ __device__ __forceinline__ double rb_pow(double p, double m) {
    if (m == 0.0) return 1.0;                    // 0 ** 0 == 1 as well
    if (!(p > 0.0)) return p == 0.0 ? 0.0 : pow(p, m);      // 0, negative (invalid particle) or NaN: the library's answer
    return fast_exp(m * fast_log(p));
}

template <> struct Model<QSMC_MODEL_RB> {
    static constexpr int D = 3;
    static __host__ __device__ __forceinline__ double lik(const double *p, const ExpArgs &e, int64_t o) {
        // rb.py:190-193: pr0 = 1 - (A * p**m + B)
        const double pr0 = 1.0 - (p[1] * rb_pow(p[0], e.m) + p[2]);
        return two_outcome(pr0, o);
    }
    static __host__ __device__ __forceinline__ bool valid(const double *p, double) {
        // rb.py:165-176
        const double P = p[0], A = p[1], B = p[2];
        return 0.0 <= P && P <= 1.0 && 0.0 <= A && A <= 1.0 && 0.0 <= B && B <= 1.0 &&
               A + B <= 1.0 && A * P + B <= 1.0;
    }
};

template <> struct Model<QSMC_MODEL_RB_INTERLEAVED> {
    static constexpr int D = 4;
    static __host__ __device__ __forceinline__ double lik(const double *p, const ExpArgs &e, int64_t o) {
        // rb.py:181-186: p_C = p_tilde * p; p = where(reference, p, p_C)
        const double pe = e.reference ? p[1] : p[0] * p[1];
        const double pr0 = 1.0 - (p[2] * rb_pow(pe, e.m) + p[3]);
        return two_outcome(pr0, o);
    }
    static __host__ __device__ __forceinline__ bool valid(const double *p, double) {
        // rb.py:150-163 (first column is named p_C there)
        const double C = p[0], P = p[1], A = p[2], B = p[3];
        return 0.0 <= P && P <= 1.0 && 0.0 <= C && C <= 1.0 && 0.0 <= A && A <= 1.0 &&
               0.0 <= B && B <= 1.0 && A + B <= 1.0 && A * P + B <= 1.0 && A * C + B <= 1.0;
    }
};

// BinomialModel over the RB models (the model simple_est_rb builds, simple_est.py:212): n_meas sequences
// of length m, `o` of them survived; pr1 = L_RB(outcome 1) = 1 - pr0.
template <> struct Model<QSMC_MODEL_BINOMIAL_RB> {
    static constexpr int D = 3;
    static __host__ __device__ __forceinline__ double lik(const double *p, const ExpArgs &e, int64_t o) {
        const double pr0 = 1.0 - (p[1] * rb_pow(p[0], e.m) + p[2]);
        return binom_pmf(1.0 - pr0, e, o);
    }
    static __host__ __device__ __forceinline__ bool valid(const double *p, double) {
        return Model<QSMC_MODEL_RB>::valid(p, 0.0);
    }
};

template <> struct Model<QSMC_MODEL_BINOMIAL_RB_INTERLEAVED> {
    static constexpr int D = 4;
    static __host__ __device__ __forceinline__ double lik(const double *p, const ExpArgs &e, int64_t o) {
        const double pe = e.reference ? p[1] : p[0] * p[1];
        const double pr0 = 1.0 - (p[2] * rb_pow(pe, e.m) + p[3]);
        return binom_pmf(1.0 - pr0, e, o);
    }
    static __host__ __device__ __forceinline__ bool valid(const double *p, double) {
        return Model<QSMC_MODEL_RB_INTERLEAVED>::valid(p, 0.0);
    }
};

template <> struct Model<QSMC_MODEL_TOMOGRAPHY> {
    static constexpr int D = QSMC_MAX_D;
    static __host__ __device__ __forceinline__ double lik(const double *p, const ExpArgs &e, int64_t o) {
        // tomography/models.py:216-226: pr1 = clip(sum_i meas_i x_i, 0, 1); pr0 = 1 - pr1
        double s = 0.0;
        for (int i = 0; i < e.d; ++i) s += e.meas[i] * p[i];
        const double pr1 = fmin(fmax(s, 0.0), 1.0);
        return two_outcome(1.0 - pr1, o);
    }
    static __host__ __device__ __forceinline__ bool valid(const double *, double) { return true; }
};

template <> struct Model<QSMC_MODEL_UNKNOWN_T2> {
    static constexpr int D = 2;
    static __host__ __device__ __forceinline__ double lik(const double *p, const ExpArgs &e, int64_t o) {
        // test_models.py:247-257: visibility = exp(-t / T2); pr0 = vis cos^2(w t / 2) + (1 - vis) / 2
        const double vis = fast_exp(-e.t * p[1]);
        const double pr0 = vis * cos_sq(p[0] * e.t / 2.0) + (1.0 - vis) / 2.0;
        return two_outcome(pr0, o);
    }
    static __host__ __device__ __forceinline__ bool valid(const double *p, double) {
        return p[0] >= 0.0 && p[1] >= 0.0;       // test_models.py:244-245
    }
};

// Likelihood as the kernels use it: the model's, raised to the MLEModel power when one is set
// (derived_models.py:689-691, `L ** self._pow`).  POW is a template parameter of the hot kernels -- an
// inlined pow() behind a run-time branch cost the fused update kernel 30 VGPRs and a wave of occupancy
// (44 -> 47 us) even when unused; the contract / design kernels take the run-time form.
template <int KIND, bool POW>
__host__ __device__ __forceinline__ double model_lik(const double *p, const ExpArgs &e, int64_t o) {
    const double L = Model<KIND>::lik(p, e, o);
    if (POW) return L > 0.0 ? fast_exp(e.lik_pow * fast_log(L)) : pow(L, e.lik_pow);     // (see rb_pow: same error argument)
    return L;
}

template <int KIND>
__host__ __device__ __forceinline__ double model_lik_rt(const double *p, const ExpArgs &e, int64_t o) {
    const double L = Model<KIND>::lik(p, e, o);
    return e.lik_pow == 0.0 ? L : pow(L, e.lik_pow);
}

// The experiment-design pass (k_hyp_sums) evaluates MANY outcomes of ONE experiment per particle.  For the binomial
// models everything but the outcome is shared: pr1 (a cos^2 or an RB survival law), and ln pr1 / ln(1 - pr1), from
// which ln pmf = ln C + k ln p + (n - k) ln(1 - p) follows with two multiply-adds per outcome.  Round 2 evaluated the
// full likelihood and a logarithm per outcome: 26 cos^2 and 26 logarithms per particle for a Binomial(25) experiment.
template <int KIND>
struct HypPre {                                           // generic: nothing shared
    __host__ __device__ __forceinline__ void prepare(const double *, const ExpArgs &) {}
    __host__ __device__ __forceinline__ void eval(const double *p, const ExpArgs &e, int64_t o, double &L, double &logL) const {
        L = model_lik_rt<KIND>(p, e, o);
        logL = L > 0.0 ? fast_log(L) : 0.0;
    }
};
struct HypPreBinomial {
    double pr1, lp, lq;
    __host__ __device__ __forceinline__ void set(double pr1_) {
        pr1 = pr1_;
        lp = pr1_ > 0.0 ? fast_log(pr1_) : 0.0;
        lq = pr1_ < 1.0 ? fast_log1m(pr1_) : 0.0;
    }
    __host__ __device__ __forceinline__ void eval(const double *, const ExpArgs &e, int64_t o, double &L, double &logL) const {
        const double pmf = binom_pmf(pr1, e, o);
        L = e.lik_pow == 0.0 ? pmf : pow(pmf, e.lik_pow);
        const double k = (double)o;
        // (0 * ln 0 never forms: a zero exponent drops its term, as in binom_pmf's log-space branch)
        const double lpm = e.log_comb + (k > 0.0 ? k * lp : 0.0) + (e.n_meas - k > 0.0 ? (e.n_meas - k) * lq : 0.0);
        logL = e.lik_pow == 0.0 ? lpm : e.lik_pow * lpm;
    }
};
template <> struct HypPre<QSMC_MODEL_BINOMIAL_PRECESSION> : HypPreBinomial {
    __host__ __device__ __forceinline__ void prepare(const double *p, const ExpArgs &e) { set(1.0 - precession_pr0(p[0], e)); }
};
template <> struct HypPre<QSMC_MODEL_BINOMIAL_RB> : HypPreBinomial {
    __host__ __device__ __forceinline__ void prepare(const double *p, const ExpArgs &e) {
        const double pr0 = 1.0 - (p[1] * rb_pow(p[0], e.m) + p[2]);
        set(1.0 - pr0);
    }
};
template <> struct HypPre<QSMC_MODEL_BINOMIAL_RB_INTERLEAVED> : HypPreBinomial {
    __host__ __device__ __forceinline__ void prepare(const double *p, const ExpArgs &e) {
        const double pe = e.reference ? p[1] : p[0] * p[1];
        const double pr0 = 1.0 - (p[2] * rb_pow(pe, e.m) + p[3]);
        set(1.0 - pr0);
    }
};

// pr1 alone (the same expressions as HypPre<KIND>::prepare), for the design kernel that needs no logarithm of it
template <int KIND> __host__ __device__ __forceinline__ double hyp_pr1(const double *p, const ExpArgs &e);
template <> __host__ __device__ __forceinline__ double hyp_pr1<QSMC_MODEL_BINOMIAL_PRECESSION>(const double *p, const ExpArgs &e) {
    return 1.0 - precession_pr0(p[0], e);
}
template <> __host__ __device__ __forceinline__ double hyp_pr1<QSMC_MODEL_BINOMIAL_RB>(const double *p, const ExpArgs &e) {
    const double pr0 = 1.0 - (p[1] * rb_pow(p[0], e.m) + p[2]);
    return 1.0 - pr0;
}
template <> __host__ __device__ __forceinline__ double hyp_pr1<QSMC_MODEL_BINOMIAL_RB_INTERLEAVED>(const double *p, const ExpArgs &e) {
    const double pe = e.reference ? p[1] : p[0] * p[1];
    const double pr0 = 1.0 - (p[2] * rb_pow(pe, e.m) + p[3]);
    return 1.0 - pr0;
}

// Runtime-dispatched validity (used by kernels that are not templated on the model).
__host__ __device__ __forceinline__ bool model_valid(int kind, const double *p, double min_freq) {
    switch (kind) {
        case QSMC_MODEL_PRECESSION:
        case QSMC_MODEL_BINOMIAL_PRECESSION: return p[0] > min_freq;
        case QSMC_MODEL_RB:
        case QSMC_MODEL_BINOMIAL_RB: return Model<QSMC_MODEL_RB>::valid(p, 0.0);
        case QSMC_MODEL_RB_INTERLEAVED:
        case QSMC_MODEL_BINOMIAL_RB_INTERLEAVED: return Model<QSMC_MODEL_RB_INTERLEAVED>::valid(p, 0.0);
        case QSMC_MODEL_UNKNOWN_T2: return Model<QSMC_MODEL_UNKNOWN_T2>::valid(p, 0.0);
        default: return true;
    }
}

// ---------------------------------------------------------------------------------------------
// Wave64 / workgroup reductions.  Deterministic: fixed shuffle tree, then wave 0..3 in order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = QSMC_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, QSMC_WAVE);
    return v;
}

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = QSMC_WAVE / 2; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, QSMC_WAVE));
    return v;
}

// Reduce K per-thread sums over the 256-thread block; thread 0 gets the totals in v[].
// lds must hold QSMC_WAVES_PER_BLOCK * K doubles.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *lds) {
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = wave_sum(v[k]);
        if (lane == 0) lds[wave * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double s = lds[k];
#pragma unroll
            for (int wv = 1; wv < QSMC_WAVES_PER_BLOCK; ++wv) s += lds[wv * K + k];
            v[k] = s;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ double block_min(double v, double *lds) {
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
    v = wave_min(v);
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int wv = 1; wv < QSMC_WAVES_PER_BLOCK; ++wv) v = fmin(v, lds[wv]);
    }
    __syncthreads();
    return v;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (Salmon et al. 2011).  One call -> 128 bits -> two 53-bit uniforms.
// Counter layout: (particle_lo, particle_hi, epoch * 2^16 + round, slot); key = seed.
// ---------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

__host__ __device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per word pair (v_mad_u64_u32) instead of separate mul_hi / mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c.z;
        c = U4{(uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// NumPy's random_sample construction: (a >> 5) * 2^26 + (b >> 6)) / 2^53  in [0, 1)
__host__ __device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// ---------------------------------------------------------------------------------------------
// Box-Muller arithmetic for the samplers.  The resampling kernel is VALU-issue bound and the library
// log / sqrt / sincospi (written for all of double's range: denormals, huge arguments, NaN/inf paths)
// were ~190 of its ~600 VALU instructions per output pair.  The arguments here live in narrow, benign
// ranges -- ln on [2^-53, 1], sqrt on [0, 74], sin/cos(pi t) on t in [0, 2) -- so plain published
// algorithms without the special-case plumbing do (~85 instructions), each good to ~1-2 ulp
// (tests/test_gpu_parity.py::test_box_muller_accuracy checks them against an 80-bit reference).
// ---------------------------------------------------------------------------------------------
// n / d for finite d > 0: v_rcp_f64 seed, two Newton steps, one residual correction
__device__ __forceinline__ double bm_div(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}

// ln x for normal x in (0, 1]: the classic argument reduction x = 2^k m, m in [sqrt(1/2), sqrt(2)),
// s = f / (2 + f) with f = m - 1, ln(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2))   (Sun fdlibm e_log.c)
__device__ __forceinline__ double bm_log(double x) {
    int hx = __double2hiint(x);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;                   // mantissa >= sqrt(2): use m / 2
    const double m = __hiloint2double(hx | (i ^ 0x3ff00000), __double2loint(x));
    k += i >> 20;
    const double f = m - 1.0;
    const double s = bm_div(f, 2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}

// sqrt x for 0 <= x < 2^100: v_rsq_f64 seed + two coupled Newton (Goldschmidt) steps + one correction
__device__ __forceinline__ double bm_sqrt(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    g = fma(fma(-g, g, x), h, g);
    return x > 0.0 ? g : 0.0;
}

// The cosine coefficients come from constant memory, i.e. scalar loads into SGPRs: as literals every Horner step
// was a v_mov_b64 (the constant into the accumulator) + v_fmac_f64, and the sampler's loop carried 63 such moves
// per output pair; as SGPR operands of v_fma_f64 they are free.  Only one of the polynomials: 16 more SGPRs fit,
// 32 (both) or the logarithm's as well spill SGPRs to VGPR lanes and cost more v_readlane than they save
// (measured: sampler 96.5 us with literals, 92.0 with these eight, 94.8 with sixteen, 97.9 with the log's too).
__constant__ double BM_COS[8] = {4.303069587032947e-06, -1.046381049248457e-04, 1.9295743094039231e-03,
                                 -2.580689139001406e-02, 2.353306303588932e-01, -1.3352627688545895,
                                 4.0587121264167685,     -4.934802200544679};
// sin(pi t), cos(pi t) for t in [0, 2]: t = q / 2 + y with q = rint(2 t) and |y| <= 1/4 (exact), Taylor
// polynomials in y with the powers of pi folded into the coefficients, quadrant fix-up by q mod 4
__device__ __forceinline__ void bm_sincospi(double t, double &sn, double &cs) {
    const double q = rint(2.0 * t);
    const double y = fma(-0.5, q, t);
    const int iq = (int)q;
    const double z = y * y;
    double ps = fma(z, -2.1915353447830217e-05, 4.6630280576761255e-04);
    ps = fma(z, ps, -7.3704309457143504e-03);
    ps = fma(z, ps, 8.214588661112823e-02);
    ps = fma(z, ps, -5.992645293207921e-01);
    ps = fma(z, ps, 2.5501640398773455);
    ps = fma(z, ps, -5.16771278004997);
    ps = fma(z, ps, 3.141592653589793);
    ps *= y;
    double pc = fma(z, BM_COS[0], BM_COS[1]);
    pc = fma(z, pc, BM_COS[2]);
    pc = fma(z, pc, BM_COS[3]);
    pc = fma(z, pc, BM_COS[4]);
    pc = fma(z, pc, BM_COS[5]);
    pc = fma(z, pc, BM_COS[6]);
    pc = fma(z, pc, BM_COS[7]);
    pc = fma(z, pc, 1.0);
    const double a = (iq & 1) ? pc : ps;                       // |sin|-like / |cos|-like by parity of q
    const double b = (iq & 1) ? ps : pc;
    sn = (iq & 2) ? -a : a;                                    // q mod 4: (s,c) = (ps,pc) (pc,-ps) (-ps,-pc) (-pc,ps)
    cs = ((iq + 1) & 2) ? -b : b;
}

struct PhiloxStream {
    uint64_t particle;
    uint32_t epoch_round;     // (epoch << 16) | round  (epoch < 65536 per seed bump; host advances seed)
    uint32_t k0, k1;
    __device__ __forceinline__ void uniforms(uint32_t slot, double &u0, double &u1) const {
        const U4 r = philox4x32_10(U4{(uint32_t)particle, (uint32_t)(particle >> 32), epoch_round, slot},
                                   k0, k1);
        u0 = u53(r.x, r.y);
        u1 = u53(r.z, r.w);
    }
    // Box-Muller pair from one Philox block.
    __device__ __forceinline__ void normals(uint32_t slot, double &z0, double &z1) const {
        double u0, u1;
        uniforms(slot, u0, u1);
        const double r = bm_sqrt(-2.0 * bm_log(1.0 - u0));  // 1 - u0 in [2^-53, 1]
        double s, c;
        bm_sincospi(2.0 * u1, s, c);
        z0 = r * c;
        z1 = r * s;
    }
};


// ---------------------------------------------------------------------------------------------
// Tomography canonicalize for ONE particle (tomography/models.py:149-209): rho = sum_a p_a B_a,
// complex-Hermitian cyclic Jacobi, clamp negative eigenvalues, re-expand, renormalise trace.
// Three pieces, so that the basis-dependent contractions can be swapped for sparse ones:
//   TomoDense<DIM>    rho <-> p through the dense (D, DIM, DIM) complex128 basis (interleaved re, im): any basis;
//   TomoPauli2        the reference's 2-qubit Pauli basis (pauli_basis(2), bases.py:137-154: tensor products of
//                     (I, X, Y, Z) / sqrt 2, first qubit slowest): every element has four non-zero entries, so rho and
//                     the re-expansion are ~50 additions each instead of 2 x 512 multiply-adds and 1024 basis loads;
//   jacobi_clamp      the eigendecomposition and the clamped reconstruction, basis-free.
// The rotation's scalars (one |h|, three divisions and two square roots per pivot in the textbook form) use the
// hardware reciprocal / reciprocal-square-root seeds with two Newton steps: a rotation that is off by an ulp is still
// unitary to an ulp, and Jacobi corrects itself in the next sweep.
// ---------------------------------------------------------------------------------------------
#ifdef __HIP_DEVICE_COMPILE__
__device__ __forceinline__ double j_rsqrt(double x) {             // x > 0, normal range
    double r = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    r = r * fma(-h * r, r, 1.5);
    r = r * fma(-h * r, r, 1.5);
    return r;
}
__device__ __forceinline__ double j_rcp(double x) {               // x != 0, normal range
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
#else
__host__ __device__ inline double j_rsqrt(double x) { return 1.0 / sqrt(x); }
__host__ __device__ inline double j_rcp(double x) { return 1.0 / x; }
#endif

// In: Hermitian A, LOWER triangle (Ar[r][c], Ai[r][c] for r >= c; the upper entries are never touched: a full iterate kept
// both mirror images live across the sweep loop -- 12 more doubles in a kernel at 200 VGPRs).  Out: any eigenvalue
// negative?  If so R = V max(lambda, 0) V^H (tomography/models.py:185-192).
template <int DIM>
__host__ __device__ inline bool jacobi_clamp(double (&Ar)[DIM][DIM], double (&Ai)[DIM][DIM], double (&Rr)[DIM][DIM],
                                             double (&Ri)[DIM][DIM]) {
    // (contraction ON in here, against the library's -ffp-contract=off: the rotations are complex multiply-adds, 986
    //  v_mul_f64 + 516 v_add_f64 per particle of k_tomo_canon_list uncontracted; nothing outside this function reproduces
    //  its intermediate bits -- the reference's eigh does not either, G5 holds the RESULT to 1e-12 -- and `on` contracts
    //  inside source expressions only, so every kernel that inlines this function gets the same fmas)
#pragma clang fp contract(on)
    // a_rc for any (r, c) from the stored triangle (indices are compile-time constants after unrolling: the branches fold)
    auto GR = [&](int r, int c) -> double { return r >= c ? Ar[r][c] : Ar[c][r]; };
    auto GI = [&](int r, int c) -> double { return r > c ? Ai[r][c] : (r == c ? 0.0 : -Ai[c][r]); };
    auto SET = [&](int r, int c, double re, double im) {
        if (r >= c) { Ar[r][c] = re; Ai[r][c] = im; }
        else { Ar[c][r] = re; Ai[c][r] = -im; }
    };
    double Vr[DIM][DIM], Vi[DIM][DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            Vr[r][c] = (r == c) ? 1.0 : 0.0;
            Vi[r][c] = 0.0;
        }
    // cyclic complex Jacobi sweeps
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0, diag2 = 0.0;
#pragma unroll
        for (int r = 0; r < DIM; ++r) diag2 += Ar[r][r] * Ar[r][r];
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
            for (int c = r + 1; c < DIM; ++c) off += Ar[c][r] * Ar[c][r] + Ai[c][r] * Ai[c][r];
        // converged: every off-diagonal entry at machine epsilon of the diagonal's norm (sum of squares <= 1e-30 diag2, i.e.
        // |a_rc| <= 1e-15 ||diag||: what eigh delivers; the iteration squares `off` per sweep, and the former 1e-34 bought a
        // whole further sweep whenever a sweep landed between the two).  Pivots already below their share of that bound
        // are not rotated.
        if (off <= 1e-30 * diag2) break;
        const double skip2 = (1e-30 / (DIM * (DIM - 1) / 2)) * diag2;
#pragma unroll
        for (int pI = 0; pI < DIM; ++pI)
#pragma unroll
            for (int q = pI + 1; q < DIM; ++q) {
                const double hr = GR(pI, q), hi = GI(pI, q);
                const double mag2 = hr * hr + hi * hi;
                if (mag2 < 1e-290 || mag2 <= skip2) continue;
                // phase e^{i phi} = h / |h|
                const double imag = j_rsqrt(mag2);
                const double er = hr * imag, ei = hi * imag;
                const double tau = (Ar[q][q] - Ar[pI][pI]) * (0.5 * imag);
                const double t2 = 1.0 + tau * tau;
                const double rt = t2 * j_rsqrt(t2);                 // sqrt(1 + tau^2)
                const double tt = (tau >= 0.0 ? 1.0 : -1.0) * j_rcp(fabs(tau) + rt);
                const double cs = j_rsqrt(1.0 + tt * tt);
                const double sn = tt * cs;
                // A <- J^H A J with J = [[c, s e^{i phi}], [-s e^{-i phi}, c]] on (p, q), using that A stays Hermitian:
                // only the entries (k, p), (k, q) of the OTHER rows are rotated (col_p' = c col_p - s conj(e) col_q,
                // col_q' = s e col_p + c col_q); the 2 x 2 pivot block has the closed form
                // a_pp' = a_pp - t |h|, a_qq' = a_qq + t |h|, a_pq' = 0 (t = tan of the rotation angle).  Round 2 rotated all
                // four rows' columns and then all four columns' rows -- 224 multiply-adds per pivot for what is 56 here;
                // with the eigenvector update (unchanged) a pivot costs ~230 instead of ~400 flops, and this kernel runs
                // at the fp64 VALU's rate.
                const double th = tt * (mag2 * imag);               // t |h|
#pragma unroll
                for (int r = 0; r < DIM; ++r) {
                    if (r != pI && r != q) {
                        const double apr = GR(r, pI), api = GI(r, pI), aqr = GR(r, q), aqi = GI(r, q);
                        const double npr = cs * apr - sn * (er * aqr + ei * aqi);
                        const double npi = cs * api - sn * (er * aqi - ei * aqr);
                        const double nqr = sn * (er * apr - ei * api) + cs * aqr;
                        const double nqi = sn * (er * api + ei * apr) + cs * aqi;
                        SET(r, pI, npr, npi);
                        SET(r, q, nqr, nqi);
                    }
                    const double vpr = Vr[r][pI], vpi = Vi[r][pI], vqr = Vr[r][q], vqi = Vi[r][q];
                    Vr[r][pI] = cs * vpr - sn * (er * vqr + ei * vqi);
                    Vi[r][pI] = cs * vpi - sn * (er * vqi - ei * vqr);
                    Vr[r][q] = sn * (er * vpr - ei * vpi) + cs * vqr;
                    Vi[r][q] = sn * (er * vpi + ei * vpr) + cs * vqi;
                }
                Ar[pI][pI] -= th;
                Ar[q][q] += th;
                Ar[q][pI] = 0.0;
                Ai[q][pI] = 0.0;
            }
    }
    bool any_neg = false;
    double lam[DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
        lam[r] = Ar[r][r];
        any_neg |= !(lam[r] >= 0.0);
    }
    if (any_neg) {
#pragma unroll
        for (int r = 0; r < DIM; ++r) lam[r] = lam[r] < 0.0 ? 0.0 : lam[r];
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                double sr = 0.0, si = 0.0;
#pragma unroll
                for (int k = 0; k < DIM; ++k) {
                    // V[r][k] * lam[k] * conj(V[c][k])
                    sr += lam[k] * (Vr[r][k] * Vr[c][k] + Vi[r][k] * Vi[c][k]);
                    si += lam[k] * (Vi[r][k] * Vr[c][k] - Vr[r][k] * Vi[c][k]);
                }
                Rr[r][c] = sr;
                Ri[r][c] = si;
            }
    }
    return any_neg;
}

// ---------------------------------------------------------------------------------------------
// Round 5: the clamped reconstruction WITHOUT eigenvectors (dim 4: the second pass of a 2-qubit canonicalize).
// jacobi_clamp spends ~45 % of a sweep on the eigenvector matrix (64 of ~146 instructions per pivot) and 512 more on
// V max(lambda, 0) V^H at the end -- to evaluate f(A) for f(x) = max(x, 0), which needs no eigenvector: f(A) is the
// polynomial of degree < DIM that interpolates f on the spectrum.  In Newton's form over the ASCENDING eigenvalues
//     f(A) = f[l1] + (A - l1) ( f[l1,l2] + (A - l2) ( f[l1,l2,l3] + (A - l3) f[l1,l2,l3,l4] ) )
// with divided differences of a function that is linear on either side of its one kink: a first-order difference is
// exactly 0 between two negative eigenvalues and exactly 1 between two positive ones, so clusters on one side of zero --
// the near-degenerate spectra that make eigenvectors (and spectral projectors) ill-defined -- cost nothing.  Horner's
// scheme in the matrix is two Hermitian products (A A and X A: polynomials in A commute and stay Hermitian, so one
// triangle each).  The eigenvalues come from the same cyclic Jacobi sweeps, minus the eigenvector update: accurate to
// eps ||A|| whatever the gaps.  ~3400 instead of ~8000 instructions per particle; result within 5e-15 of the
// eigenvector form over 6e5 Ginibre-plus-noise matrices and over prescribed spectra with gaps 1e-1 ... 1e-15 (two or
// three negative eigenvalues clustered, a negative next to a positive, all four together, all four around zero) --
// EXCEPT three or more eigenvalues clustered ACROSS zero (a nearly pure state seen through noise): the second-order
// differences of the kink over a cluster of width g are O(1 / g) and the result is off by ~4e-15 / g.  Those are flagged
// (return value 2: any three consecutive eigenvalues that straddle zero within 1e-2 ||A||_F) and the caller sends them
// through jacobi_clamp; on a Ginibre-like cloud that is never.
// Returns 0: no negative eigenvalue (R not formed); 1: R = f(A), lower triangle; 2: flagged, R not valid.
// ---------------------------------------------------------------------------------------------
// Step 1: ascending eigenvalues of the Hermitian A (lower triangle; DESTROYED) and ||A||_F^2, by jacobi_clamp's sweeps
// without the eigenvector matrix.
__host__ __device__ inline void herm4_eigenvalues(double (&Ar)[4][4], double (&Ai)[4][4], double (&lam)[4], double &frob2) {
#pragma clang fp contract(on)                             // (as in jacobi_clamp: nothing reproduces these intermediates)
    constexpr int DIM = 4;
    auto GR = [&](int r, int c) -> double { return r >= c ? Ar[r][c] : Ar[c][r]; };
    auto GI = [&](int r, int c) -> double { return r > c ? Ai[r][c] : (r == c ? 0.0 : -Ai[c][r]); };
    auto SET = [&](int r, int c, double re, double im) {
        if (r >= c) { Ar[r][c] = re; Ai[r][c] = im; }
        else { Ar[c][r] = re; Ai[c][r] = -im; }
    };
    frob2 = 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0, diag2 = 0.0;
#pragma unroll
        for (int r = 0; r < DIM; ++r) diag2 += Ar[r][r] * Ar[r][r];
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
            for (int c = r + 1; c < DIM; ++c) off += Ar[c][r] * Ar[c][r] + Ai[c][r] * Ai[c][r];
        frob2 = diag2 + 2.0 * off;
        if (off <= 1e-30 * diag2) break;                  // (jacobi_clamp's bound and pivot rule: the same eigenvalues)
        const double skip2 = (1e-30 / (DIM * (DIM - 1) / 2)) * diag2;
#pragma unroll
        for (int pI = 0; pI < DIM; ++pI)
#pragma unroll
            for (int q = pI + 1; q < DIM; ++q) {
                const double hr = GR(pI, q), hi = GI(pI, q);
                const double mag2 = hr * hr + hi * hi;
                if (mag2 < 1e-290 || mag2 <= skip2) continue;
                const double imag = j_rsqrt(mag2);
                const double er = hr * imag, ei = hi * imag;
                const double tau = (Ar[q][q] - Ar[pI][pI]) * (0.5 * imag);
                const double t2 = 1.0 + tau * tau;
                const double rt = t2 * j_rsqrt(t2);
                const double tt = (tau >= 0.0 ? 1.0 : -1.0) * j_rcp(fabs(tau) + rt);
                const double cs = j_rsqrt(1.0 + tt * tt);
                const double sn = tt * cs;
                const double th = tt * (mag2 * imag);
#pragma unroll
                for (int r = 0; r < DIM; ++r) {
                    if (r != pI && r != q) {
                        const double apr = GR(r, pI), api = GI(r, pI), aqr = GR(r, q), aqi = GI(r, q);
                        SET(r, pI, cs * apr - sn * (er * aqr + ei * aqi), cs * api - sn * (er * aqi - ei * aqr));
                        SET(r, q, sn * (er * apr - ei * api) + cs * aqr, sn * (er * api + ei * apr) + cs * aqi);
                    }
                }
                Ar[pI][pI] -= th;
                Ar[q][q] += th;
                Ar[q][pI] = 0.0;
                Ai[q][pI] = 0.0;
            }
    }
    // ascending (a five-exchange network)
    double l0 = Ar[0][0], l1 = Ar[1][1], l2 = Ar[2][2], l3 = Ar[3][3];
    auto cx = [](double &a, double &b) { const double lo = fmin(a, b), hi = fmax(a, b); a = lo; b = hi; };
    cx(l0, l1); cx(l2, l3); cx(l0, l2); cx(l1, l3); cx(l1, l2);
    lam[0] = l0; lam[1] = l1; lam[2] = l2; lam[3] = l3;
}

// Step 2: what to do with a particle whose rho has these eigenvalues.  0: none negative; 2: three consecutive ones across
// zero inside 1e-2 ||A||_F (the eigenvector form's); 1: psd_from_eigenvalues4.
__host__ __device__ inline int psd_verdict4(const double (&lam)[4], double frob2) {
    const double l0 = lam[0], l1 = lam[1], l2 = lam[2], l3 = lam[3];
    if (!(l0 < 0.0)) return (l0 == l0) ? 0 : 2;          // (a NaN anywhere: let the eigenvector form decide as before)
    const double width = 1e-2 * sqrt(frob2);
    if ((l2 > 0.0 && l2 - l0 < width) || (l1 < 0.0 && l3 > 0.0 && l3 - l1 < width)) return 2;     // (l0 < 0 here)
    return 1;
}

// Step 3: R = f(A), f(x) = max(x, 0), as Newton's interpolation polynomial over the ascending eigenvalues lam (lam[0] < 0)
// evaluated at the matrix A (lower triangle in, lower triangle out).
__host__ __device__ inline void psd_from_eigenvalues4(const double (&Ar0)[4][4], const double (&Ai0)[4][4], const double (&lam)[4],
                                                      double (&Rr)[4][4], double (&Ri)[4][4]) {
#pragma clang fp contract(on)
    constexpr int DIM = 4;
    const double l0 = lam[0], l1 = lam[1], l2 = lam[2], l3 = lam[3];
    const double f0 = 0.0, f1 = fmax(l1, 0.0), f2 = fmax(l2, 0.0), f3 = fmax(l3, 0.0);
    // first differences: exactly 0 / 1 on one side of the kink (also for coinciding eigenvalues)
    auto d1 = [](double fa, double fb, double la, double lb) -> double {
        if (!(lb > 0.0)) return 0.0;
        if (!(la < 0.0)) return 1.0;
        return (fb - fa) * j_rcp(lb - la);               // la < 0 < lb
    };
    auto dn = [](double a, double b, double la, double lb) -> double {
        const double den = lb - la;
        return den > 0.0 ? (b - a) * j_rcp(den) : 0.0;
    };
    const double f01 = d1(f0, f1, l0, l1), f12 = d1(f1, f2, l1, l2), f23 = d1(f2, f3, l2, l3);
    const double f012 = dn(f01, f12, l0, l2), f123 = dn(f12, f23, l1, l3);
    const double f0123 = dn(f012, f123, l0, l3);
    // X = alpha A^2 + beta A + gamma I  =  (f0123 (A - l2) + f012) (A - l1) + f01;   R = X (A - l0) + f0 = X A - l0 X
    const double alpha = f0123, b1 = f012 - f0123 * l2;
    const double beta = b1 - alpha * l1, gamma = f01 - b1 * l1;
    auto AR = [&](int r, int c) -> double { return r >= c ? Ar0[r][c] : Ar0[c][r]; };
    auto AI = [&](int r, int c) -> double { return r > c ? Ai0[r][c] : (r == c ? 0.0 : -Ai0[c][r]); };
    double Xr[DIM][DIM], Xi[DIM][DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                sr += AR(r, k) * AR(k, c) - AI(r, k) * AI(k, c);
                if (r != c) si += AR(r, k) * AI(k, c) + AI(r, k) * AR(k, c);
            }
            Xr[r][c] = alpha * sr + beta * Ar0[r][c] + (r == c ? gamma : 0.0);
            Xi[r][c] = (r == c) ? 0.0 : alpha * si + beta * Ai0[r][c];
        }
    auto XR = [&](int r, int c) -> double { return r >= c ? Xr[r][c] : Xr[c][r]; };
    auto XI = [&](int r, int c) -> double { return r > c ? Xi[r][c] : (r == c ? 0.0 : -Xi[c][r]); };
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                sr += XR(r, k) * AR(k, c) - XI(r, k) * AI(k, c);
                if (r != c) si += XR(r, k) * AI(k, c) + XI(r, k) * AR(k, c);
            }
            Rr[r][c] = sr - l0 * Xr[r][c];
            Ri[r][c] = (r == c) ? 0.0 : si - l0 * Xi[r][c];
        }
}

// The three steps on one matrix (lower triangle in; R: lower triangle).  Returns the verdict of step 2.
__host__ __device__ inline int psd_project4(const double (&Ar0)[4][4], const double (&Ai0)[4][4], double (&Rr)[4][4],
                                            double (&Ri)[4][4]) {
    double Ar[4][4], Ai[4][4], lam[4], frob2;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) { Ar[r][c] = Ar0[r][c]; Ai[r][c] = Ai0[r][c]; }
    herm4_eigenvalues(Ar, Ai, lam, frob2);
    const int verdict = psd_verdict4(lam, frob2);
    if (verdict == 1) psd_from_eigenvalues4(Ar0, Ai0, lam, Rr, Ri);
    return verdict;
}

template <int DIM>
struct TomoDense {
    static constexpr bool NEEDS_FULL = true;              // expand() contracts with every entry of R
    const double *__restrict__ basis;
    // rho = sum_a x_a B_a; `lower_only`: the classification needs one triangle
    __host__ __device__ inline void build(const double *p, double (&Ar)[DIM][DIM], double (&Ai)[DIM][DIM], bool lower_only) const {
        constexpr int D = DIM * DIM;
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                if (lower_only && c > r) continue;
                double sr = 0.0, si = 0.0;
                for (int a = 0; a < D; ++a) {
                    sr += p[a] * basis[2 * ((a * DIM + r) * DIM + c)];
                    si += p[a] * basis[2 * ((a * DIM + r) * DIM + c) + 1];
                }
                Ar[r][c] = sr;
                Ai[r][c] = si;
            }
    }
    // x_a = Re tr(B_a^H R) = Re sum_{rc} conj(B_a[r][c]) R[r][c]
    __host__ __device__ inline void expand(const double (&Rr)[DIM][DIM], const double (&Ri)[DIM][DIM], double *p) const {
        constexpr int D = DIM * DIM;
        for (int a = 0; a < D; ++a) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < DIM; ++r)
#pragma unroll
                for (int c = 0; c < DIM; ++c)
                    s += basis[2 * ((a * DIM + r) * DIM + c)] * Rr[r][c] + basis[2 * ((a * DIM + r) * DIM + c) + 1] * Ri[r][c];
            p[a] = s;
        }
    }
};

// B_{4 i + j} = sigma_i (x) sigma_j / 2.  With M_i = sum_j x_{4 i + j} sigma_j = [[a_i, b_i - i c_i], [b_i + i c_i, d_i]]
// (a_i = x_{i0} + x_{i3}, d_i = x_{i0} - x_{i3}, b_i = x_{i1}, c_i = x_{i2}) the 2 x 2 blocks of rho are
//   R00 = (M_0 + M_3) / 2,  R11 = (M_0 - M_3) / 2,  R10 = (M_1 + i M_2) / 2,  R01 = R10^H,
// and back: with t_0(G) = g00 + g11, t_3 = g00 - g11, t_1 = g01 + g10, t_2 = i (g01 - g10) for a block G,
//   x_{0j} = Re[t_j(R00) + t_j(R11)] / 2,  x_{3j} = Re[t_j(R00) - t_j(R11)] / 2,  x_{1j} = Re t_j(R10),  x_{2j} = Im t_j(R10).
struct TomoPauli2 {
    static constexpr bool NEEDS_FULL = false;             // expand() reads the lower triangle of R
    __host__ __device__ inline void build(const double *x, double (&Ar)[4][4], double (&Ai)[4][4], bool) const {
        double a[4], dd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = x[4 * i] + x[4 * i + 3];
            dd[i] = x[4 * i] - x[4 * i + 3];
        }
        const double *b0 = x, *b1 = x + 4, *b2 = x + 8, *b3 = x + 12;      // b_i = x[4 i + 1], c_i = x[4 i + 2]
        Ar[0][0] = 0.5 * (a[0] + a[3]);            Ai[0][0] = 0.0;
        Ar[1][1] = 0.5 * (dd[0] + dd[3]);          Ai[1][1] = 0.0;
        Ar[2][2] = 0.5 * (a[0] - a[3]);            Ai[2][2] = 0.0;
        Ar[3][3] = 0.5 * (dd[0] - dd[3]);          Ai[3][3] = 0.0;
        Ar[1][0] = 0.5 * (b0[1] + b3[1]);          Ai[1][0] = 0.5 * (b0[2] + b3[2]);
        Ar[3][2] = 0.5 * (b0[1] - b3[1]);          Ai[3][2] = 0.5 * (b0[2] - b3[2]);
        Ar[2][0] = 0.5 * a[1];                     Ai[2][0] = 0.5 * a[2];
        Ar[2][1] = 0.5 * (b1[1] + b2[2]);          Ai[2][1] = 0.5 * (b2[1] - b1[2]);
        Ar[3][0] = 0.5 * (b1[1] - b2[2]);          Ai[3][0] = 0.5 * (b1[2] + b2[1]);
        Ar[3][1] = 0.5 * dd[1];                    Ai[3][1] = 0.5 * dd[2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = r + 1; c < 4; ++c) {
                Ar[r][c] = Ar[c][r];
                Ai[r][c] = -Ai[c][r];
            }
    }
    __host__ __device__ inline void expand(const double (&Rr)[4][4], const double (&Ri)[4][4], double *x) const {
        // diagonal blocks (Hermitian): t_0 = g00 + g11, t_3 = g00 - g11, t_1 = 2 Re g10, t_2 = 2 Im g10
        const double u0 = Rr[0][0] + Rr[1][1], u3 = Rr[0][0] - Rr[1][1], u1 = 2.0 * Rr[1][0], u2 = 2.0 * Ri[1][0];
        const double v0 = Rr[2][2] + Rr[3][3], v3 = Rr[2][2] - Rr[3][3], v1 = 2.0 * Rr[3][2], v2 = 2.0 * Ri[3][2];
        x[0] = 0.5 * (u0 + v0);  x[1] = 0.5 * (u1 + v1);  x[2] = 0.5 * (u2 + v2);  x[3] = 0.5 * (u3 + v3);
        x[12] = 0.5 * (u0 - v0); x[13] = 0.5 * (u1 - v1); x[14] = 0.5 * (u2 - v2); x[15] = 0.5 * (u3 - v3);
        // off-diagonal block G = R10 = [[rho20, rho21], [rho30, rho31]]
        const double g00r = Rr[2][0], g00i = Ri[2][0], g01r = Rr[2][1], g01i = Ri[2][1];
        const double g10r = Rr[3][0], g10i = Ri[3][0], g11r = Rr[3][1], g11i = Ri[3][1];
        x[4] = g00r + g11r;          x[8] = g00i + g11i;             // t_0
        x[7] = g00r - g11r;          x[11] = g00i - g11i;            // t_3
        x[5] = g01r + g10r;          x[9] = g01i + g10i;             // t_1
        x[6] = -(g01i - g10i);       x[10] = g01r - g10r;            // t_2 = i (g01 - g10): Re = -(Im g01 - Im g10), Im = Re g01 - Re g10
    }
};

// Returns true if p[] was modified.  `reload(p)` fetches p again where it is still needed after the eigendecomposition
// (no negative eigenvalue: only the trace renormalisation is left) -- the D coordinates do not stay live across the
// Jacobi sweeps then (32 VGPRs at D = 16); a no-op functor keeps them.
struct TomoKeepP { __host__ __device__ inline void operator()(double *) const {} };
template <int DIM, class Basis, class Reload = TomoKeepP>
__host__ __device__ inline bool tomo_canon_particle(const Basis &B, double *p, bool allow_subnormalized, Reload reload = Reload{}) {
    constexpr int D = DIM * DIM;
    double Ar[DIM][DIM], Ai[DIM][DIM], Rr[DIM][DIM], Ri[DIM][DIM];
    B.build(p, Ar, Ai, true);                             // (jacobi_clamp works on the lower triangle)
    const bool any_neg = jacobi_clamp<DIM>(Ar, Ai, Rr, Ri);
    if (any_neg) B.expand(Rr, Ri, p);
    else reload(p);
    if (!allow_subnormalized) {                           // :194-209 (x / (x_0 sqrt(dim)): one reciprocal, D products)
        const double inv = 1.0 / (p[0] * sqrt((double)DIM));
#pragma unroll
        for (int a = 0; a < D; ++a) p[a] = p[a] * inv;
    }
    return any_neg || !allow_subnormalized;
}

// tomo_canon_particle for dim 4 through psd_project4.  Returns 0: p untouched; 1: p rewritten; 2: flagged for the
// eigenvector form (p untouched: the caller lists the particle for tomo_canon_particle).
template <class Basis, class Reload = TomoKeepP>
__host__ __device__ inline int tomo_canon_particle4_fast(const Basis &B, double *p, bool allow_subnormalized, Reload reload = Reload{}) {
    double Ar[4][4], Ai[4][4], lam[4], frob2;
    B.build(p, Ar, Ai, true);
    herm4_eigenvalues(Ar, Ai, lam, frob2);                // (destroys the iterate; p need not stay live across the sweeps)
    const int verdict = psd_verdict4(lam, frob2);
    if (verdict == 2) return 2;
    reload(p);
    if (verdict == 1) {
        double Rr[4][4], Ri[4][4];
        B.build(p, Ar, Ai, true);                         // rho once more, for the polynomial
        psd_from_eigenvalues4(Ar, Ai, lam, Rr, Ri);
        if (Basis::NEEDS_FULL) {                          // (TomoPauli2 reads the lower triangle only)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = r + 1; c < 4; ++c) { Rr[r][c] = Rr[c][r]; Ri[r][c] = -Ri[c][r]; }
        }
        B.expand(Rr, Ri, p);
    }
    if (!allow_subnormalized) {                           // tomography/models.py:194-209
        const double inv = 1.0 / (p[0] * sqrt(4.0));
#pragma unroll
        for (int a = 0; a < 16; ++a) p[a] = p[a] * inv;
    }
    return (verdict == 1 || !allow_subnormalized) ? 1 : 0;
}

// Cheap sufficient test for "rho is positive definite": the LDL^H factorisation of the Hermitian rho
// (built from p as in tomo_canon_particle) has only positive pivots.  Used to sort a cloud into the
// particles canonicalize leaves alone (apart from the trace renormalisation) and the ones that need the
// eigendecomposition; anything not clearly positive definite (a zero or negative pivot) goes to the latter.
template <int DIM, class Basis>
__host__ __device__ inline bool tomo_clearly_positive(const Basis &B, const double *p) {
#pragma clang fp contract(on)                             // (as in jacobi_clamp: a verdict, not a reproduced value)
    double Ar[DIM][DIM], Ai[DIM][DIM];
    B.build(p, Ar, Ai, true);                             // lower triangle is enough
    // in place: A[j][j] <- d_j, A[i][j] <- l_ij (i > j)
    bool ok = true;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        double dj = Ar[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= (Ar[j][k] * Ar[j][k] + Ai[j][k] * Ai[j][k]) * Ar[k][k];
        ok = ok && (dj > 0.0);
        Ar[j][j] = dj;
        const double inv = 1.0 / dj;                     // (garbage if !ok: the verdict is already false)
#pragma unroll
        for (int i = j + 1; i < DIM; ++i) {
            double sr = Ar[i][j], si = Ai[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) {
                // l_ik * conj(l_jk) * d_k
                const double tr = Ar[i][k] * Ar[j][k] + Ai[i][k] * Ai[j][k];
                const double ti = Ai[i][k] * Ar[j][k] - Ar[i][k] * Ai[j][k];
                sr -= tr * Ar[k][k];
                si -= ti * Ar[k][k];
            }
            Ar[i][j] = sr * inv;
            Ai[i][j] = si * inv;
        }
    }
    return ok;
}

}  // namespace qsmc
