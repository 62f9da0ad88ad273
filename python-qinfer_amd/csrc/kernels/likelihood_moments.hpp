// kernels/likelihood_moments.hpp -- contract likelihood / validity kernels and the weighted moments (VALU and MFMA forms).
// Part of the single translation unit qsmc_kernels.hip (included there, in this order; not a stand-alone header).
#pragma once

// =============================================================================================
// contract likelihood / validity
// =============================================================================================
template <int KIND>
__global__ __launch_bounds__(QSMC_BLOCK) void k_likelihood(const double *__restrict__ x, int64_t ldx,
                                                           int64_t n, ExpArgs e, int64_t outcome,
                                                           double *__restrict__ L) {
    constexpr int D = Model<KIND>::D;
    const int d = (KIND == QSMC_MODEL_TOMOGRAPHY) ? e.d : D;
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double p[D];
#pragma unroll
        for (int m = 0; m < D; ++m)
            if (m < d) p[m] = x[m * ldx + i];
        L[i] = model_lik_rt<KIND>(p, e, outcome);
    }
}

__global__ __launch_bounds__(QSMC_BLOCK) void k_valid(const double *__restrict__ x, int64_t ldx,
                                                      int64_t n, int kind, int d, double min_freq,
                                                      uint8_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double p[4] = {0, 0, 0, 0};
        const int dd = d < 4 ? d : 4;
        for (int m = 0; m < dd; ++m) p[m] = x[m * ldx + i];
        out[i] = model_valid(kind, p, min_freq) ? 1 : 0;
    }
}

__global__ __launch_bounds__(QSMC_BLOCK) void k_fill(double *__restrict__ w, int64_t n, double v) {
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK)
        w[i] = v;
}

// =============================================================================================
// weighted moments:  [sum w, sum w x_m, sum w x_m x_n (m <= n)]
// =============================================================================================
template <int D>
__global__ __launch_bounds__(QSMC_BLOCK) void k_moments_small(const double *__restrict__ x, int64_t ldx,
                                                              int64_t n, const double *__restrict__ w,
                                                              double norm, ReduceOut ro) {
    constexpr int K = 1 + D + D * (D + 1) / 2;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        const double wi = w[i] / norm;
        double p[D];
#pragma unroll
        for (int m = 0; m < D; ++m) p[m] = x[m * ldx + i];
        acc[0] += wi;
        int k = 1 + D;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const double wx = wi * p[m];
            acc[1 + m] += wx;
#pragma unroll
            for (int q = m; q < D; ++q) acc[k++] += wx * p[q];
        }
    }
    block_publish<K>(acc, 0.0, ro);
}

// d in 5..16 on the matrix cores: sum_p w_p x_p x_p^T is the (d x N)(N x d) contraction X diag(w) X^T,
// the one genuinely GEMM-shaped op on this path.  v_mfma_f64_16x16x4_f64: A is 16x4, B is 4x16, lane l
// holds A[l & 15][l >> 4] and B[l >> 4][l & 15] -- with rows = parameters and the 4 k-slots =
// particles, the A and B operands of a lane are the SAME x value (times w for A), so every lane reads
// one double4 of its row (4 consecutive particles) and feeds 4 MFMAs.  C/D layout (cdna guide 3):
// value r of lane l is C[(l >> 4) + 4 r][l & 15].  First moments and sum w ride along on the VALU.
// The contraction itself is ~8 us of MFMA time at N = 1.25e6; the kernel is HBM-bound (x read once).
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int MFMA_MOM_K = 16 * 16 + 16 + 1;        // [C (256, row-major full), sum w x (16), sum w]

__global__ __launch_bounds__(QSMC_BLOCK) void k_moments_mfma(const double *__restrict__ x, int64_t ldx,
                                                             int64_t n, int d, const double *__restrict__ w,
                                                             double norm, double *__restrict__ partials) {
    __shared__ double lds[QSMC_WAVES_PER_BLOCK * MFMA_MOM_K];
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
    const int m = lane & 15, kq = lane >> 4;
    const bool row_ok = m < d;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    double s1 = 0.0, s0 = 0.0;
    // 64 particles per wave iteration (4 MFMA k-groups of 16): all 8 double4 loads of a lane are issued
    // before the first MFMA, so a wave keeps 16 rows x 512 B in flight
    const int64_t tiles = (n + 63) / 64;
    const int64_t wave_id = (int64_t)blockIdx.x * QSMC_WAVES_PER_BLOCK + wave;
    const int64_t n_waves = (int64_t)gridDim.x * QSMC_WAVES_PER_BLOCK;
    const bool vec_ok = (ldx & 3) == 0 && (((uintptr_t)x | (uintptr_t)w) & 31) == 0;
    const double inv_norm = 1.0 / norm;
    const double *xrow = x + (row_ok ? m : 0) * ldx;
    for (int64_t tile = wave_id; tile < tiles; tile += n_waves) {
        const int64_t base = tile * 64 + 4 * kq;                // + 16 g + q
        double xv[4][4], wv[4][4];
        if (vec_ok && tile * 64 + 64 <= n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double4 xx = *reinterpret_cast<const double4 *>(xrow + base + 16 * g);
                xv[g][0] = xx.x; xv[g][1] = xx.y; xv[g][2] = xx.z; xv[g][3] = xx.w;
                if (w) {
                    const double4 ww = *reinterpret_cast<const double4 *>(w + base + 16 * g);
                    wv[g][0] = ww.x; wv[g][1] = ww.y; wv[g][2] = ww.z; wv[g][3] = ww.w;
                } else {
                    wv[g][0] = wv[g][1] = wv[g][2] = wv[g][3] = 1.0;
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t p = base + 16 * g + q;
                    const bool ok = p < n;
                    wv[g][q] = ok ? (w ? w[p] : 1.0) : 0.0;
                    xv[g][q] = ok ? xrow[p] : 0.0;
                }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double wq = wv[g][q] * inv_norm;
                const double xq = row_ok ? xv[g][q] : 0.0;
                const double a = wq * xq;
                if (q & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xq, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xq, acc0, 0, 0, 0);
                s1 += a;
                s0 += wq;
            }
    }
    // first moments: lanes with the same m (4 of them, one per kq) hold partial sums
    s1 += __shfl_xor(s1, 16, QSMC_WAVE);
    s1 += __shfl_xor(s1, 32, QSMC_WAVE);
    s0 += __shfl_xor(s0, 16, QSMC_WAVE);                        // every m carries the same w sums
    s0 += __shfl_xor(s0, 32, QSMC_WAVE);
    double *mine = lds + wave * MFMA_MOM_K;
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[((kq + 4 * r) * 16) + m] = acc0[r] + acc1[r];
    if (kq == 0) mine[256 + m] = s1;
    if (lane == 0) mine[272] = s0;
    __syncthreads();
    for (int k = threadIdx.x; k < MFMA_MOM_K; k += QSMC_BLOCK) {
        double t = lds[k];
#pragma unroll
        for (int wv2 = 1; wv2 < QSMC_WAVES_PER_BLOCK; ++wv2) t += lds[wv2 * MFMA_MOM_K + k];
        partials[(size_t)blockIdx.x * MFMA_MOM_K + k] = t;
    }
}

// out[k] = sum_g partials[g * K + k], summed in g order by thread k's ... (one block, K <= 256)
__global__ __launch_bounds__(QSMC_BLOCK) void k_sum_partials(const double *__restrict__ partials,
                                                             int nblocks, int K, double *__restrict__ out) {
    // each wave handles a set of k; lanes stride over g; fixed shuffle tree -> deterministic
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
    for (int k = blockIdx.x * QSMC_WAVES_PER_BLOCK + wave; k < K; k += gridDim.x * QSMC_WAVES_PER_BLOCK) {
        double s = 0.0;
        for (int g = lane; g < nblocks; g += QSMC_WAVE) s += partials[(size_t)g * K + k];
        s = wave_sum(s);
        if (lane == 0) out[k] = s;
    }
}

// mapped[k] = sum_g partials[k * nblocks + g] (column-major rows of block_publish) for k < K, in pinned host memory, and the
// completion word behind them, in one launch: a wave per column, lanes stride over the rows (coalesced), fixed shuffle tree
// -> deterministic; every workgroup fences its pinned stores at system scope and draws a ticket, the last one sets the
// word (the ticket resets itself).  The design kernel's passes (qsmc_hypothetical_sums_*: one such launch per pass, each
// publishing to its own slot).  A handle is bound to ONE stream at a time (include/qsmc.h): the single ticket word relies
// on launches from that stream being serialised.
__global__ __launch_bounds__(QSMC_BLOCK) void k_sum_columns_publish(const double *__restrict__ partials, int nblocks, int K,
                                                                    double *__restrict__ mapped, unsigned long long *flag,
                                                                    unsigned long long seq, unsigned int *ticket) {
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
    for (int k = blockIdx.x * QSMC_WAVES_PER_BLOCK + wave; k < K; k += gridDim.x * QSMC_WAVES_PER_BLOCK) {
        double s = 0.0;
        for (int g = lane; g < nblocks; g += QSMC_WAVE) s += partials[(size_t)k * nblocks + g];
        s = wave_sum(s);
        if (lane == 0) {
            mapped[k] = s;
            __threadfence_system();                          // (the writing lane only: a fence in every thread cost 7 us)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (relaxed: an agent-scope RELEASE here writes back the XCD's whole L2 -- measured: the merged kernel at 14.5 us against
        //  7.0 + 4.7 for the two it replaced; what must be ordered before the ticket is in pinned memory and already fenced)
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1u) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            *reinterpret_cast<volatile unsigned long long *>(flag) = seq;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel-density cross term of est_kl_divergence (distributions.py:466-487; distances metrics.py:72-106):
//     sum_i p_i log( sum_j q_j phi(|| sqrt(Q) (x_i - y_j) ||_2 / delta) ),   phi = standard normal pdf,
// p = w / norm_p over the n particles x, q = v / norm_q over the m particles y (null weights: all ones).
// O(n m d): the reference forms the n x m distance matrix on the host; here a thread owns one x_i and the y_j go
// past in LDS tiles of 256 (every lane reads the same y_j: an LDS broadcast), nothing of size n m exists.
// `scale` = sqrt(Q) / delta per parameter, applied to both clouds before the subtraction.
// ---------------------------------------------------------------------------------------------
constexpr int KDE_TILE = QSMC_BLOCK;
struct KdeScale { double s[QSMC_MAX_D]; };

__global__ __launch_bounds__(QSMC_BLOCK) void k_kde_cross(const double *__restrict__ x, int64_t ldx, int64_t n,
                                                          const double *__restrict__ w, double inv_norm_p,
                                                          const double *__restrict__ y, int64_t ldy, int64_t m,
                                                          const double *__restrict__ v, double inv_norm_q, int d,
                                                          KdeScale sc, ReduceOut ro) {
    __shared__ double ys[QSMC_MAX_D][KDE_TILE];
    __shared__ double vs[KDE_TILE];
    double total[3] = {0.0, 0.0, 0.0};
    for (int64_t base = (int64_t)blockIdx.x * QSMC_BLOCK; base < n; base += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t i = base + threadIdx.x;
        const bool live = i < n;
        double xi[QSMC_MAX_D];
#pragma unroll
        for (int q = 0; q < QSMC_MAX_D; ++q) xi[q] = (q < d && live) ? x[q * ldx + i] * sc.s[q] : 0.0;
        double acc = 0.0;
        for (int64_t j0 = 0; j0 < m; j0 += KDE_TILE) {
            __syncthreads();
            const int64_t j = j0 + threadIdx.x;
            for (int q = 0; q < d; ++q) ys[q][threadIdx.x] = j < m ? y[q * ldy + j] * sc.s[q] : 0.0;
            vs[threadIdx.x] = j < m ? (v ? v[j] : 1.0) * inv_norm_q : 0.0;
            __syncthreads();
            const int len = (int)((m - j0) < KDE_TILE ? (m - j0) : KDE_TILE);
            for (int jj = 0; jj < len; ++jj) {
                double z2 = 0.0;
#pragma unroll
                for (int q = 0; q < QSMC_MAX_D; ++q)
                    if (q < d) {
                        const double t = xi[q] - ys[q][jj];
                        z2 += t * t;
                    }
                acc += vs[jj] * exp(-0.5 * z2);
            }
        }
        if (live) {
            const double dens = acc * 0.3989422804014327;                       // 1 / sqrt(2 pi)
            total[0] += ((w ? w[i] : 1.0) * inv_norm_p) * log(dens);            // log 0 = -inf like the reference
        }
    }
    block_publish<3>(total, INFINITY, ro);
}

