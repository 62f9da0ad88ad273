// kernels/sqrtm.hpp -- utils.sqrtm_psd (reference utils.py:593-607) as ONE algorithm with two executions: a host loop
// (qsmc_sqrtm_psd) and a single wavefront on the device (lw_sqrt16_wave, d = 16), the same IEEE operations in the same
// order, so the two return the same bits.  Part of the single translation unit qsmc_kernels.hip.
#pragma once

// The reference takes eigh of the covariance, clamps negative eigenvalues to 0 and returns (v sqrt(w)) v^H with the
// Frobenius error || S S - A ||.  Here: cyclic two-sided Jacobi in ROUND-ROBIN order.  A sweep of an n x n matrix
// (m = n rounded up to even) is m - 1 rounds of m / 2 index pairs that are disjoint within a round (the circle
// method: position 0 stays, the others rotate by one per round).  Rotations on disjoint pairs commute exactly in
// their parameters -- (a_pp, a_qq, a_pq) of one pair is not touched by the rotation of another -- so a round is:
//   1. every pair's (c, s) from the matrix as the round finds it;
//   2. all column rotations of A and of V;
//   3. all row rotations of A.
// Which is what lets one wavefront take a round in three steps instead of 8 x 3 dependent ones: a serial cyclic sweep
// of a 16 x 16 matrix is 120 dependent rotations (each a division, two square roots and another two divisions deep),
// ~0.4 us apiece on one lane.  Round 3's host routine swept row by row (p, q > p); the order changed here, on both
// sides, and with it the last bits of S (both orders converge to the same matrix within rounding).
// Convergence test (per sweep): off = sum_i (sum_{j > i} a_ij^2), diag = sum_i a_ii^2, both in index order.

// element at position j of the circle in round r (m positions, position 0 fixed)
__host__ __device__ __forceinline__ int rr_elem(int r, int j, int m) {
    if (j == 0) return 0;
    int e = (j - 1 - r) % (m - 1);
    if (e < 0) e += m - 1;
    return e + 1;
}

__host__ __device__ __forceinline__ void jacobi_cs(double app, double aqq, double apq, double &c, double &s) {
    const double tau = (aqq - app) / (2.0 * apq);
    const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    c = 1.0 / sqrt(1.0 + t * t);
    s = t * c;
}

constexpr int SQRTM_MAX_SWEEPS = 64;
constexpr double SQRTM_OFF_TOL = 1e-34;

// Host execution, n <= 64.  A row-major n x n; S_out = scale * sqrt(A); *err_out = || sqrt(A) sqrt(A) - A ||_F.
// `a`, `v`, `sq`: caller-provided work arrays of n * n doubles each (sqrtm_psd_host below sizes them by n: the step
// path's d <= 16 on the stack -- 6 KB -- anything larger from the heap; a fixed 3 x 64 x 64 = 96 KB of stack was
// within reach of a small-stack worker thread calling the C ABI).
static void sqrtm_psd_host_work(const double *A, int n, double scale, double *S_out, double *err_out,
                                double *lambda_min_out, double *a, double *v, double *sq) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            a[i * n + j] = 0.5 * (A[i * n + j] + A[j * n + i]);   // eigh reads one triangle; symmetrise
            v[i * n + j] = (i == j) ? 1.0 : 0.0;
        }
    const int m = n + (n & 1);
    for (int sweep = 0; sweep < SQRTM_MAX_SWEEPS && n > 1; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            double row = 0.0;
            for (int j = i + 1; j < n; ++j) row += a[i * n + j] * a[i * n + j];
            off += row;
            diag += a[i * n + i] * a[i * n + i];
        }
        if (off == 0.0 || off <= SQRTM_OFF_TOL * diag) break;
        for (int r = 0; r < m - 1; ++r) {
            int P[32], Q[32], act[32];
            double C[32], Sn[32];
            for (int i = 0; i < m / 2; ++i) {
                const int e0 = rr_elem(r, i, m), e1 = rr_elem(r, m - 1 - i, m);
                const int p = e0 < e1 ? e0 : e1, q = e0 < e1 ? e1 : e0;
                P[i] = p;
                Q[i] = q;
                act[i] = q < n && a[p * n + q] != 0.0;
                if (act[i]) jacobi_cs(a[p * n + p], a[q * n + q], a[p * n + q], C[i], Sn[i]);
            }
            for (int i = 0; i < m / 2; ++i) {
                if (!act[i]) continue;
                const int p = P[i], q = Q[i];
                const double c = C[i], s = Sn[i];
                for (int k = 0; k < n; ++k) {
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq;
                    a[k * n + q] = s * akp + c * akq;
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq;
                    v[k * n + q] = s * vkp + c * vkq;
                }
            }
            for (int i = 0; i < m / 2; ++i) {
                if (!act[i]) continue;
                const int p = P[i], q = Q[i];
                const double c = C[i], s = Sn[i];
                for (int k = 0; k < n; ++k) {
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk;
                    a[q * n + k] = s * apk + c * aqk;
                }
            }
        }
    }
    if (lambda_min_out) {
        double mn = a[0];
        for (int k = 1; k < n; ++k) mn = a[k * n + k] < mn ? a[k * n + k] : mn;
        *lambda_min_out = mn;
    }
    // S = V sqrt(max(lambda, 0)) V^T
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) {
                const double lam = a[k * n + k];
                const double rt = lam <= 0.0 ? 0.0 : sqrt(lam);
                s += v[i * n + k] * rt * v[j * n + k];
            }
            sq[i * n + j] = s;
        }
    if (err_out) {
        double e2 = 0.0;
        for (int i = 0; i < n; ++i) {
            double row = 0.0;
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += sq[i * n + k] * sq[k * n + j];
                const double dlt = s - A[i * n + j];
                row += dlt * dlt;
            }
            e2 += row;
        }
        *err_out = sqrt(e2);
    }
    for (int k = 0; k < n * n; ++k) S_out[k] = scale * sq[k];
}

// Returns false when the work arrays of an n > 16 matrix could not be allocated (the caller reports QSMC_ERR_ALLOC).
static bool sqrtm_psd_host(const double *A, int n, double scale, double *S_out, double *err_out,
                           double *lambda_min_out = nullptr) {
    constexpr int SMALL = QSMC_MAX_D;
    if (n <= SMALL) {
        double a[SMALL * SMALL], v[SMALL * SMALL], sq[SMALL * SMALL];
        sqrtm_psd_host_work(A, n, scale, S_out, err_out, lambda_min_out, a, v, sq);
        return true;
    }
    double *work = static_cast<double *>(malloc(sizeof(double) * 3 * (size_t)n * (size_t)n));
    if (!work) return false;
    sqrtm_psd_host_work(A, n, scale, S_out, err_out, lambda_min_out, work, work + (size_t)n * n, work + 2 * (size_t)n * n);
    free(work);
    return true;
}

// ---------------------------------------------------------------------------------------------
// Device execution for the d = 16 Liu-West resample: the work qsmc_step did on the host between k_moments_mfma and
// the kick kernel (resamplers.py:266-300) -- mean = sum w x, cov = sum w x x^T - mean mean^T, the zero-covariance
// substitute, S = h sqrtm_psd(cov) with its error -- by ONE wavefront, from the summed moments in device memory to the
// LWDev block the kick kernel reads.  The host receives moments, S and the error through pinned memory, runs
// sqrtm_psd_host on the same covariance and adopts the queued resample only if every bit agrees (qsmc_step).
// ---------------------------------------------------------------------------------------------
struct LWDev {               // device-resident Liu-West arguments of a d = 16 resample (written by lw_sqrt16_wave)
    double a;
    double mean[QSMC_MAX_D];
    double S[QSMC_MAX_D * QSMC_MAX_D];
    double valid;            // 1: use; 0: covariance not finite / square root error not finite -- the kick kernel leaves at once
};

struct SqrtJob {             // by-value argument of the kernel that carries the wavefront (k_bucket_anc16); full == nullptr: none
    const double *full;      // [MFMA_MOM_K] summed moments: C (256, row-major), sum w x (16), sum w -- weights already / norm
    LWDev *out;
    double *mapped;          // pinned host block (device alias): [0, 273) moments, [288, 544) S, [544] error, [545] valid
    unsigned long long *flag;
    unsigned long long seq;
    double a, h, zero_cov_comp;
};
constexpr int SQRT_MAPPED_S = 288, SQRT_MAPPED_ERR = 544, SQRT_MAPPED_VALID = 545, SQRT_MAPPED_DOUBLES = 1024;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One wavefront (64 lanes; the caller guarantees threadIdx.x < 64 and that no other wave touches `lds`): n = 16.
// lds: at least 16 * 17 * 3 + 64 doubles.
__device__ __forceinline__ void lw_sqrt16_wave(const SqrtJob job, double *lds) {
    constexpr int N = 16, LD = 17;
    double *A = lds, *V = lds + N * LD, *A0 = lds + 2 * N * LD, *tmp = lds + 3 * N * LD;   // tmp[64]
    const int lane = threadIdx.x & 63;
    const int row = lane & 15, quad = lane >> 4;
    const double *hf = job.full;
    // publish the moments (what k_publish_big did) and form mean / covariance exactly as the host does from them
    for (int k = lane; k < MFMA_MOM_K; k += 64) job.mapped[k] = hf[k];
    double mean_r = hf[256 + row];
    bool finite = true, any = false;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = quad * 4 + t;
        // (row <= col reads C[row][col], else C[col][row]: the host fills both halves from the packed upper triangle)
        const double e2 = row <= col ? hf[row * 16 + col] : hf[col * 16 + row];
        const double mc = hf[256 + col];
        const double cv = e2 - mean_r * mc;
        A0[row * LD + col] = cv;
        finite = finite && isfinite(cv);
        any = any || cv != 0.0;
    }
    finite = __all(finite);
    any = __any(any);
    if (!any) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = quad * 4 + t;
            A0[row * LD + col] = row == col ? job.zero_cov_comp : 0.0;
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = quad * 4 + t;
        A[row * LD + col] = 0.5 * (A0[row * LD + col] + A0[col * LD + row]);
        V[row * LD + col] = row == col ? 1.0 : 0.0;
    }
    wave_lds_sync();
    if (finite) {
        for (int sweep = 0; sweep < SQRTM_MAX_SWEEPS; ++sweep) {
            // convergence test: row sums by lanes 0..15, totals by every lane in index order (all lanes agree)
            if (lane < N) {
                double rs = 0.0;
                for (int j = lane + 1; j < N; ++j) rs += A[lane * LD + j] * A[lane * LD + j];
                tmp[lane] = rs;
                tmp[16 + lane] = A[lane * LD + lane] * A[lane * LD + lane];
            }
            wave_lds_sync();
            double off = 0.0, diag = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                off += tmp[i];
                diag += tmp[16 + i];
            }
            wave_lds_sync();                     // (tmp is rewritten below)
            if (off == 0.0 || off <= SQRTM_OFF_TOL * diag) break;
            for (int r = 0; r < N - 1; ++r) {
                if (lane < N / 2) {
                    const int e0 = rr_elem(r, lane, N), e1 = rr_elem(r, N - 1 - lane, N);
                    const int p = e0 < e1 ? e0 : e1, q = e0 < e1 ? e1 : e0;
                    const double apq = A[p * LD + q];
                    double c = 1.0, s = 0.0;
                    const bool act = apq != 0.0;
                    if (act) jacobi_cs(A[p * LD + p], A[q * LD + q], apq, c, s);
                    tmp[4 * lane] = c;
                    tmp[4 * lane + 1] = s;
                    tmp[4 * lane + 2] = act ? (double)(p * 16 + q) : -1.0;
                }
                wave_lds_sync();
                // columns of A and V: lane (row, quad) takes pairs quad and quad + 4
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int i = quad + 4 * t;
                    const double code = tmp[4 * i + 2];
                    if (code >= 0.0) {
                        const int pq = (int)code, p = pq >> 4, q = pq & 15;
                        const double c = tmp[4 * i], s = tmp[4 * i + 1];
                        const double akp = A[row * LD + p], akq = A[row * LD + q];
                        A[row * LD + p] = c * akp - s * akq;
                        A[row * LD + q] = s * akp + c * akq;
                        const double vkp = V[row * LD + p], vkq = V[row * LD + q];
                        V[row * LD + p] = c * vkp - s * vkq;
                        V[row * LD + q] = s * vkp + c * vkq;
                    }
                }
                wave_lds_sync();
                // rows of A
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int i = quad + 4 * t;
                    const double code = tmp[4 * i + 2];
                    if (code >= 0.0) {
                        const int pq = (int)code, p = pq >> 4, q = pq & 15;
                        const double c = tmp[4 * i], s = tmp[4 * i + 1];
                        const double apk = A[p * LD + row], aqk = A[q * LD + row];
                        A[p * LD + row] = c * apk - s * aqk;
                        A[q * LD + row] = s * apk + c * aqk;
                    }
                }
                wave_lds_sync();
            }
        }
    }
    // sq = V sqrt(max(lambda, 0)) V^T -> A (its diagonal is read first), error against A0, S = h sq
    double rt[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double lam = A[k * LD + k];
        rt[k] = lam <= 0.0 ? 0.0 : sqrt(lam);
    }
    wave_lds_sync();
    double sqv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = quad * 4 + t;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) s += V[row * LD + k] * rt[k] * V[col * LD + k];
        sqv[t] = s;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) A[row * LD + quad * 4 + t] = sqv[t];
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = quad * 4 + t;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) s += A[row * LD + k] * A[k * LD + col];
        const double dlt = s - A0[row * LD + col];
        V[row * LD + col] = dlt * dlt;           // (V is free now)
    }
    wave_lds_sync();
    if (lane < N) {
        double rs = 0.0;
        for (int j = 0; j < N; ++j) rs += V[lane * LD + j];
        tmp[lane] = rs;
    }
    wave_lds_sync();
    double e2 = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) e2 += tmp[i];
    const double err = sqrt(e2);
    const double valid = (finite && isfinite(err)) ? 1.0 : 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = quad * 4 + t;
        const double sv = job.h * sqv[t];
        job.out->S[row * 16 + col] = sv;
        job.mapped[SQRT_MAPPED_S + row * 16 + col] = sv;
    }
    if (lane < N) job.out->mean[lane] = hf[256 + lane];
    if (lane == 0) {
        job.out->a = job.a;
        job.out->valid = valid;
        job.mapped[SQRT_MAPPED_ERR] = err;
        job.mapped[SQRT_MAPPED_VALID] = valid;
    }
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) *reinterpret_cast<volatile unsigned long long *>(job.flag) = job.seq;
}
