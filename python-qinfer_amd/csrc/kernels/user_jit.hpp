// kernels/user_jit.hpp -- the fused update for a USER model, compiled at run time (round 6).
//
// The reference's plugin contract is `Model.likelihood(outcomes, modelparams, expparams)` (abstract_model.py:444-468): any
// user class, NumPy in and out.  Served literally, that is a whole-cloud D2H, a host evaluation and an upload per datum
// (150 ms at N = 1e7); with the torch hook (`likelihood_device`) it is eight eager elementwise kernels and a weight pass
// (0.52 ms).  The library's own models run ONE pass of 16 + 8 d bytes per particle because their likelihood is inlined into
// the update kernel.  This file gives a user model the same: the model states its per-particle likelihood as a HIP device
// function (`likelihood_hip`, a string), hiprtc compiles it INTO the kernel below for gfx950, and the update is one launch
// with the reductions the rest of the path expects -- partial rows in block_publish's layout ([NS + 1][grid], column-major:
// [sum w', sum w'^2, #bad, sum w' x (d), upper(sum w' x x^T), min w']), finished by the library's own k_reduce_partials.
//
// What the user source must define (QSMC_D = n_modelparams and QSMC_NEP = number of experiment doubles are predefined):
//     __device__ double likelihood(const double *x, const double *ep, long long outcome);
//         x[0 .. QSMC_D): one particle;  ep[0 .. QSMC_NEP): the experiment's record fields as doubles, in dtype order
//         (vector fields flattened);  returns Pr(outcome | x; ep).
// and may define, announcing it with `#define QSMC_USER_HAS_VALID 1`:
//     __device__ bool valid(const double *x);            // are_models_valid (abstract_model.py:286-300)
#pragma once

static const char *const USER_JIT_PRELUDE = R"JIT(
#define QSMC_DMOM (QSMC_D <= 4 ? QSMC_D : 0)
#define QSMC_NS (3 + QSMC_DMOM + QSMC_DMOM * (QSMC_DMOM + 1) / 2)
#define QSMC_JIT_BLOCK 256
#define QSMC_JIT_UNROLL (QSMC_D <= 2 ? 4 : (QSMC_D <= 4 ? 2 : 1))      /* pairs of particles a thread holds per tile */
)JIT";

static const char *const USER_JIT_KERNELS = R"JIT(
#ifndef QSMC_USER_HAS_VALID
#define QSMC_USER_HAS_VALID 0
#endif
struct QsmcUserEp { double v[QSMC_NEP > 0 ? QSMC_NEP : 1]; };

__device__ inline double qsmc_wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ inline double qsmc_wave_min(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}

// w_out[i] = (w_in[i] / prev_norm) * likelihood(x_i)   (w_in == nullptr: all-ones weights), and the update's sums.
// A workgroup's tile is 256 x 2 x QSMC_JIT_UNROLL particles; vec != 0 (every pointer 16-byte aligned, ldx even): a full tile
// is read with 16-byte loads, ALL of them issued before the first likelihood is evaluated (as k_update_fused does); ragged
// tiles and unaligned clouds take the guarded scalar path.
struct QsmcJitAcc {
    double s[QSMC_NS];
    double mn;
    __device__ inline void add(double w, const double *p) {
        s[0] += w;
        s[1] += w * w;
        s[2] += (w >= 0.0) ? 0.0 : 1.0;
        mn = fmin(mn, w);
        int k = 3 + QSMC_DMOM;
#pragma unroll
        for (int m = 0; m < QSMC_DMOM; ++m) {
            const double wx = w * p[m];
            s[3 + m] += wx;
#pragma unroll
            for (int q = m; q < QSMC_DMOM; ++q) s[k++] += wx * p[q];
        }
    }
};

extern "C" __global__ __launch_bounds__(QSMC_JIT_BLOCK) void qsmc_user_update(
    const double *__restrict__ x, long long ldx, long long n, const double *__restrict__ w_in,
    double *__restrict__ w_out, double prev_norm, QsmcUserEp ep, long long outcome, double *__restrict__ partials, int vec) {
    __shared__ double lds[(QSMC_JIT_BLOCK / 64) * (QSMC_NS + 1)];
    QsmcJitAcc acc;
#pragma unroll
    for (int k = 0; k < QSMC_NS; ++k) acc.s[k] = 0.0;
    acc.mn = __builtin_huge_val();
    const double inv_norm = 1.0 / prev_norm;
    const long long tile = (long long)QSMC_JIT_BLOCK * 2 * QSMC_JIT_UNROLL;
    for (long long base = (long long)blockIdx.x * tile; base < n; base += (long long)gridDim.x * tile) {
        if (vec && base + tile <= n) {
            double2 wv[QSMC_JIT_UNROLL], xv[QSMC_JIT_UNROLL][QSMC_D];
#pragma unroll
            for (int u = 0; u < QSMC_JIT_UNROLL; ++u) {
                const long long i2 = (base >> 1) + (long long)u * QSMC_JIT_BLOCK + threadIdx.x;       // in pairs
                wv[u] = w_in ? reinterpret_cast<const double2 *>(w_in)[i2] : make_double2(1.0, 1.0);
#pragma unroll
                for (int m = 0; m < QSMC_D; ++m) xv[u][m] = reinterpret_cast<const double2 *>(x + (long long)m * ldx)[i2];
            }
#pragma unroll
            for (int u = 0; u < QSMC_JIT_UNROLL; ++u) {
                const long long i2 = (base >> 1) + (long long)u * QSMC_JIT_BLOCK + threadIdx.x;
                double pa[QSMC_D], pb[QSMC_D];
#pragma unroll
                for (int m = 0; m < QSMC_D; ++m) { pa[m] = xv[u][m].x; pb[m] = xv[u][m].y; }
                const double wa = (wv[u].x * inv_norm) * likelihood(pa, ep.v, outcome);
                const double wb = (wv[u].y * inv_norm) * likelihood(pb, ep.v, outcome);
                reinterpret_cast<double2 *>(w_out)[i2] = make_double2(wa, wb);
                acc.add(wa, pa);
                acc.add(wb, pb);
            }
        } else {
            for (int u = 0; u < 2 * QSMC_JIT_UNROLL; ++u) {
                const long long i = base + (long long)u * QSMC_JIT_BLOCK + threadIdx.x;
                if (i < n) {
                    double p[QSMC_D];
#pragma unroll
                    for (int m = 0; m < QSMC_D; ++m) p[m] = x[(long long)m * ldx + i];
                    const double w = ((w_in ? w_in[i] : 1.0) * inv_norm) * likelihood(p, ep.v, outcome);
                    w_out[i] = w;
                    acc.add(w, p);
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < QSMC_NS; ++k) acc.s[k] = qsmc_wave_sum(acc.s[k]);
    acc.mn = qsmc_wave_min(acc.mn);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < QSMC_NS; ++k) lds[wave * (QSMC_NS + 1) + k] = acc.s[k];
        lds[wave * (QSMC_NS + 1) + QSMC_NS] = acc.mn;
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= QSMC_NS; k += QSMC_JIT_BLOCK) {
        double t = lds[k];
        for (int wv2 = 1; wv2 < QSMC_JIT_BLOCK / 64; ++wv2) {
            const double o = lds[wv2 * (QSMC_NS + 1) + k];
            t = (k < QSMC_NS) ? t + o : fmin(t, o);
        }
        partials[(unsigned long long)k * gridDim.x + blockIdx.x] = t;
    }
}

// A batch_update window (smc.py:459-487; k_update_multi's contract): k <= 8 data in ONE pass, w_j = w_{j-1} L_j without
// renormalisation, per-datum sums [S_j, Q_j, #bad_j] at s[3 j ..], the moments of the final weights behind them, the
// window's minimum last -- the layout qsmc_update_multi's host side reads.
struct QsmcUserWindow {
    double ep[8][QSMC_NEP > 0 ? QSMC_NEP : 1];
    long long outcome[8];
    int k;
};
#define QSMC_NSW (24 + QSMC_DMOM + QSMC_DMOM * (QSMC_DMOM + 1) / 2)
extern "C" __global__ __launch_bounds__(QSMC_JIT_BLOCK) void qsmc_user_update_multi(
    const double *__restrict__ x, long long ldx, long long n, const double *__restrict__ w_in,
    double *__restrict__ w_out, double prev_norm, QsmcUserWindow win, double *__restrict__ partials) {
    __shared__ double lds[(QSMC_JIT_BLOCK / 64) * (QSMC_NSW + 1)];
    double s[QSMC_NSW];
#pragma unroll
    for (int q = 0; q < QSMC_NSW; ++q) s[q] = 0.0;
    double mn = __builtin_huge_val();
    const double inv_norm = 1.0 / prev_norm;
    const long long tile = (long long)QSMC_JIT_BLOCK * 8;
    for (long long base = (long long)blockIdx.x * tile; base < n; base += (long long)gridDim.x * tile) {
#pragma unroll 1
        for (int u = 0; u < 8; ++u) {
            const long long i = base + (long long)u * QSMC_JIT_BLOCK + threadIdx.x;
            if (i < n) {
                double p[QSMC_D];
#pragma unroll
                for (int m = 0; m < QSMC_D; ++m) p[m] = x[(long long)m * ldx + i];
                double w = (w_in ? w_in[i] : 1.0) * inv_norm;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < win.k) {
                        w = w * likelihood(p, win.ep[j], win.outcome[j]);
                        s[3 * j] += w;
                        s[3 * j + 1] += w * w;
                        s[3 * j + 2] += (w >= 0.0) ? 0.0 : 1.0;
                        mn = fmin(mn, w);
                    }
                }
                w_out[i] = w;
                int q = 24 + QSMC_DMOM;
#pragma unroll
                for (int m = 0; m < QSMC_DMOM; ++m) {
                    const double wx = w * p[m];
                    s[24 + m] += wx;
#pragma unroll
                    for (int m2 = m; m2 < QSMC_DMOM; ++m2) s[q++] += wx * p[m2];
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < QSMC_NSW; ++q) s[q] = qsmc_wave_sum(s[q]);
    mn = qsmc_wave_min(mn);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < QSMC_NSW; ++q) lds[wave * (QSMC_NSW + 1) + q] = s[q];
        lds[wave * (QSMC_NSW + 1) + QSMC_NSW] = mn;
    }
    __syncthreads();
    for (int q = threadIdx.x; q <= QSMC_NSW; q += QSMC_JIT_BLOCK) {
        double t = lds[q];
        for (int wv2 = 1; wv2 < QSMC_JIT_BLOCK / 64; ++wv2) {
            const double o = lds[wv2 * (QSMC_NSW + 1) + q];
            t = (q < QSMC_NSW) ? t + o : fmin(t, o);
        }
        partials[(unsigned long long)q * gridDim.x + blockIdx.x] = t;
    }
}

// L_out[i] = likelihood(x_i) for one (outcome, experiment) pair
extern "C" __global__ __launch_bounds__(QSMC_JIT_BLOCK) void qsmc_user_likelihood(
    const double *__restrict__ x, long long ldx, long long n, QsmcUserEp ep, long long outcome, double *__restrict__ L_out) {
    for (long long i = (long long)blockIdx.x * QSMC_JIT_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * QSMC_JIT_BLOCK) {
        double p[QSMC_D];
#pragma unroll
        for (int m = 0; m < QSMC_D; ++m) p[m] = x[(long long)m * ldx + i];
        L_out[i] = likelihood(p, ep.v, outcome);
    }
}

extern "C" __global__ __launch_bounds__(QSMC_JIT_BLOCK) void qsmc_user_valid(
    const double *__restrict__ x, long long ldx, long long n, unsigned char *__restrict__ mask) {
    for (long long i = (long long)blockIdx.x * QSMC_JIT_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * QSMC_JIT_BLOCK) {
#if QSMC_USER_HAS_VALID
        double p[QSMC_D];
#pragma unroll
        for (int m = 0; m < QSMC_D; ++m) p[m] = x[(long long)m * ldx + i];
        mask[i] = valid(p) ? 1 : 0;
#else
        mask[i] = 1;
#endif
    }
}
)JIT";
