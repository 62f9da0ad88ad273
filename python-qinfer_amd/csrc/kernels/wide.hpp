// kernels/wide.hpp -- clouds of 16 < d <= 64 parameters: tomography beyond two qubits (dim 5 .. 8, d = dim^2; three qubits:
// dim 8, d = 64).  The reference's TomographyModel takes any dim (tomography/models.py:82-226); rounds 1-5 stopped at
// QSMC_MAX_D = 16 because the narrow kernels carry a particle (and S, the mean, the measurement vector) in registers /
// the kernarg segment.  Here nothing of size d lives in a lane across a loop and nothing of size d^2 rides in a kernarg:
//   k_update_tomo_wide / k_likelihood_wide   Pr(1 | x) = clip(meas . x, 0, 1) over the NONZERO entries of meas (a Pauli
//                         measurement touches 2 of 64 rows: 16 + 8 nnz bytes per particle), rows streamed one at a time;
//   k_moments_wide<NB>    sum w x x^T as NB (NB + 1) / 2 blocks of 16 x 16 on v_mfma_f64_16x16x4 (NB = ceil(d / 16));
//   k_anc_direct + k_kick_wide<NB>   Liu-West: ancestors (small clouds; large ones reuse k_bucket_anc16, which needs the
//                         weights only) and the kicks S z as NB x NB chains of the same MFMA, S and the mean from device memory;
//   k_centres_wide / k_perturb_wide   the legacy-RNG pieces (resamplers.py:318-372) for d > 16;
//   k_tomo_classify_wide<DIM> / k_tomo_canon_list_wide<DIM>   canonicalize (tomography/models.py:149-209), dim 5 .. 8.
// Part of the single translation unit qsmc_kernels.hip (included there after resample.hpp and walk_tomo.hpp).
#pragma once

constexpr int WIDE_D = QSMC_MAX_D_WIDE;
struct TomoWideArgs {
    int32_t d, nnz;
    double lik_pow;              // MLEModel power (0 = plain)
    int32_t idx[WIDE_D];         // the rows meas does not vanish on, ascending
    double val[WIDE_D];          // ... and its entries there
};

__device__ __forceinline__ double tomo_wide_lik(double s, int64_t outcome, double lik_pow) {
    // tomography/models.py:216-226: pr1 = clip(sum_i meas_i x_i, 0, 1); outcome 0 -> 1 - pr1
    const double pr1 = fmin(fmax(s, 0.0), 1.0);
    const double L = two_outcome(1.0 - pr1, outcome);
    return lik_pow == 0.0 ? L : pow(L, lik_pow);
}

// Same tiles, tile sums and partial rows as k_update_tomo (update.hpp): everything behind it (reduction, chunk prefix,
// speculative counts, resample prefix) is unchanged.  The sum runs over the nonzero entries in ascending row order -- the
// skipped terms are +-0 -- in separate multiplies and adds (-ffp-contract=off), like the narrow kernels.
template <int VEC, bool ONES, bool NT>     // NT: streaming hints on the row loads (a pass beyond the Infinity Cache)
__global__ __launch_bounds__(QSMC_BLOCK) void k_update_tomo_wide(
    const double *__restrict__ x, int64_t ldx, int64_t n, const double *__restrict__ w_in,
    double *__restrict__ w_out, double prev_norm, TomoWideArgs e, int64_t outcome, ReduceOut ro) {
    constexpr int64_t TILE = (int64_t)QSMC_BLOCK * VEC * UPD_UNROLL;
    typedef double nt2 __attribute__((ext_vector_type(2)));
    UpdAcc<0> acc;
    acc.init();
    const double inv_norm = 1.0 / prev_norm;
    for (int64_t base = (int64_t)blockIdx.x * TILE; base < n; base += (int64_t)gridDim.x * TILE) {
        double tsum = 0.0;
        if (VEC == 2 && base + TILE <= n) {
            double s0[UPD_UNROLL], s1[UPD_UNROLL];
#pragma unroll
            for (int u = 0; u < UPD_UNROLL; ++u) { s0[u] = 0.0; s1[u] = 0.0; }
#pragma unroll 2                                             // (4: 127 -> 141 us for a dense vector at N = 1e6)
            for (int j = 0; j < e.nnz; ++j) {                  // (uniform: idx / val come out of the kernarg segment)
                const double *row = x + (int64_t)e.idx[j] * ldx + base;
                const double mv = e.val[j];
#pragma unroll
                for (int u = 0; u < UPD_UNROLL; ++u) {
                    double2 xv;
                    if (NT) {               // (a dense vector at N = 1e6 reads 528 MB: 127 -> 85 us with the hint; inside the cache it costs)
                        const nt2 t = __builtin_nontemporal_load(reinterpret_cast<const nt2 *>(row + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2));
                        xv.x = t.x;
                        xv.y = t.y;
                    } else {
                        xv = *reinterpret_cast<const double2 *>(row + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2);
                    }
                    s0[u] += mv * xv.x;
                    s1[u] += mv * xv.y;
                }
            }
#pragma unroll
            for (int u = 0; u < UPD_UNROLL; ++u) {
                const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2;
                double2 wi;
                if (ONES) { wi.x = 1.0; wi.y = 1.0; } else wi = *reinterpret_cast<const double2 *>(w_in + i);
                double2 wo;
                wo.x = (wi.x * inv_norm) * tomo_wide_lik(s0[u], outcome, e.lik_pow);
                wo.y = (wi.y * inv_norm) * tomo_wide_lik(s1[u], outcome, e.lik_pow);
                *reinterpret_cast<double2 *>(w_out + i) = wo;
                acc.add(wo.x, nullptr);
                acc.add(wo.y, nullptr);
                tsum += wo.x + wo.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < UPD_UNROLL; ++u) {
#pragma unroll
                for (int hh = 0; hh < VEC; ++hh) {
                    const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * VEC + hh;
                    if (i < n) {
                        double s = 0.0;
                        for (int j = 0; j < e.nnz; ++j) s += e.val[j] * x[(int64_t)e.idx[j] * ldx + i];
                        const double wo = ((ONES ? 1.0 : w_in[i]) * inv_norm) * tomo_wide_lik(s, outcome, e.lik_pow);
                        w_out[i] = wo;
                        acc.add(wo, nullptr);
                        tsum += wo;
                    }
                }
            }
        }
        if (ro.tile_sums) {                      // uniform
            const double t = wave_sum(tsum);
            if ((threadIdx.x & (QSMC_WAVE - 1)) == 0)
                ro.tile_sums[(base / TILE) * QSMC_WAVES_PER_BLOCK + threadIdx.x / QSMC_WAVE] = t;
        }
    }
    if (ro.tile_sums) {                          // (as in k_update_fused: zero the last chunk's missing tiles)
        static_assert(4096 % TILE == 0, "tiles per chunk");
        constexpr int64_t PER_CHUNK = 4096 / TILE * QSMC_WAVES_PER_BLOCK;
        const int64_t last = (n - 1) / TILE;
        if ((int64_t)blockIdx.x == last % (int64_t)gridDim.x) {
            const int64_t first = (last + 1) * QSMC_WAVES_PER_BLOCK;
            const int64_t end = (first + PER_CHUNK - 1) / PER_CHUNK * PER_CHUNK;
            for (int64_t k = first + threadIdx.x; k < end; k += QSMC_BLOCK) ro.tile_sums[k] = 0.0;
        }
    }
    block_publish<3>(acc.s, acc.mn, ro);
}

__global__ __launch_bounds__(QSMC_BLOCK) void k_likelihood_wide(const double *__restrict__ x, int64_t ldx, int64_t n,
                                                                TomoWideArgs e, int64_t outcome, double *__restrict__ L) {
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double s = 0.0;
        for (int j = 0; j < e.nnz; ++j) s += e.val[j] * x[(int64_t)e.idx[j] * ldx + i];
        L[i] = tomo_wide_lik(s, outcome, e.lik_pow);
    }
}

// =============================================================================================
// weighted moments, 16 < d <= 64: X diag(w) X^T in 16 x 16 blocks (upper block triangle) on the f64 matrix cores.
// Operand layout as in k_moments_mfma: lane l = (m = l & 15, kq = l >> 4) holds A[m][kq] and B[kq][m] of one MFMA step;
// with rows = parameters and the 4 k-slots = particles both are x values of the lane's own row -- for block pair
// (bi, bj): A = w x_{16 bi + m}, B = x_{16 bj + m} of particle 4 kq + q (step q).  A lane reads one double4 (4 consecutive
// particles) of its row in each of the NB row blocks per 16 particles and feeds 4 NB (NB + 1) / 2 MFMAs.
// d = 64: 40 MFMAs (64 cycles each at 78.6 TFLOP/s) per 2.1 KB read -- 65 us of matrix time per 1e6 particles against 66 us
// of HBM time for the 528 B per particle: balanced.  Partial sums leave a workgroup block pair by block pair through 8 KB of LDS.
// Per-workgroup row: [pair 0: 256 (row-major 16 x 16) | pair 1 | ... | sum w x (16 NB) | sum w].
// =============================================================================================
constexpr int wide_pairs(int nb) { return nb * (nb + 1) / 2; }
constexpr int wide_mom_k(int nb) { return wide_pairs(nb) * 256 + 16 * nb + 1; }

template <int NB>
__global__ __launch_bounds__(QSMC_BLOCK, 3) void k_moments_wide(const double *__restrict__ x, int64_t ldx, int64_t n, int d,
                                                             const double *__restrict__ w, double norm,
                                                             double *__restrict__ partials, int nt) {
    constexpr int NP = wide_pairs(NB), K = wide_mom_k(NB);
    __shared__ double lds[QSMC_WAVES_PER_BLOCK * 256];
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    const int m = lane & 15, kq = lane >> 4;
    v4d acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = v4d{0.0, 0.0, 0.0, 0.0};
    double s1[NB], s0 = 0.0;
    bool row_ok[NB];
    const double *xrow[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        s1[b] = 0.0;
        row_ok[b] = 16 * b + m < d;
        xrow[b] = x + (int64_t)(row_ok[b] ? 16 * b + m : 0) * ldx;
    }
    const int64_t tiles = (n + 15) / 16;
    const int64_t wave_id = (int64_t)blockIdx.x * QSMC_WAVES_PER_BLOCK + wave;
    const int64_t n_waves = (int64_t)gridDim.x * QSMC_WAVES_PER_BLOCK;
    const bool vec_ok = (ldx & 3) == 0 && (((uintptr_t)x | (uintptr_t)w) & 31) == 0;
    const double inv_norm = 1.0 / norm;
    for (int64_t tile = wave_id; tile < tiles; tile += n_waves) {
        const int64_t base = tile * 16 + 4 * kq;
        double xv[NB][4], wv[4];
        if (vec_ok && tile * 16 + 16 <= n) {
            if (w) {
                const double4 ww = *reinterpret_cast<const double4 *>(w + base);
                wv[0] = ww.x; wv[1] = ww.y; wv[2] = ww.z; wv[3] = ww.w;
            } else {
                wv[0] = wv[1] = wv[2] = wv[3] = 1.0;
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (nt) {                   // (uniform) a cloud beyond the Infinity Cache: streaming hint (dense update: 127 -> 85 us)
                    const v4d xx = __builtin_nontemporal_load(reinterpret_cast<const v4d *>(xrow[b] + base));
                    xv[b][0] = xx[0]; xv[b][1] = xx[1]; xv[b][2] = xx[2]; xv[b][3] = xx[3];
                } else {
                    const double4 xx = *reinterpret_cast<const double4 *>(xrow[b] + base);
                    xv[b][0] = xx.x; xv[b][1] = xx.y; xv[b][2] = xx.z; xv[b][3] = xx.w;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t p = base + q;
                const bool ok = p < n;
                wv[q] = ok ? (w ? w[p] : 1.0) : 0.0;
#pragma unroll
                for (int b = 0; b < NB; ++b) xv[b][q] = ok ? xrow[b][p] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double wq = wv[q] * inv_norm;
            double a[NB], xq[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                xq[b] = row_ok[b] ? xv[b][q] : 0.0;
                a[b] = wq * xq[b];
                s1[b] += a[b];
            }
            s0 += wq;
            int p = 0;
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int bj = bi; bj < NB; ++bj) {
                    acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], xq[bj], acc[p], 0, 0, 0);
                    ++p;
                }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        s1[b] += __shfl_xor(s1[b], 16, QSMC_WAVE);
        s1[b] += __shfl_xor(s1[b], 32, QSMC_WAVE);
    }
    s0 += __shfl_xor(s0, 16, QSMC_WAVE);
    s0 += __shfl_xor(s0, 32, QSMC_WAVE);
    double *row_out = partials + (size_t)blockIdx.x * K;
    double *mine = lds + wave * 256;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        // C/D layout: value r of lane l = C[(l >> 4) + 4 r][l & 15]
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(kq + 4 * r) * 16 + m] = acc[p][r];
        __syncthreads();
        {
            const int k = threadIdx.x;                      // QSMC_BLOCK == 256 entries of the block
            double t = lds[k];
#pragma unroll
            for (int wv2 = 1; wv2 < QSMC_WAVES_PER_BLOCK; ++wv2) t += lds[wv2 * 256 + k];
            row_out[p * 256 + k] = t;
        }
        __syncthreads();
    }
    // first moments and sum w: through the same LDS block
    if (kq == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) mine[16 * b + m] = s1[b];
    }
    if (lane == 0) mine[16 * NB] = s0;
    __syncthreads();
    if (threadIdx.x <= 16 * NB) {
        const int k = threadIdx.x;
        double t = lds[k];
#pragma unroll
        for (int wv2 = 1; wv2 < QSMC_WAVES_PER_BLOCK; ++wv2) t += lds[wv2 * 256 + k];
        row_out[NP * 256 + k] = t;
    }
}

// =============================================================================================
// Liu-West for 16 < d <= 64.  a, the mean and S = h sqrtm(cov) come from device memory (LWWide: 33 KB do not fit a kernarg
// segment), uploaded by the host call; rows and columns beyond d are zero.
// =============================================================================================
struct LWWide {
    double a, pad[3];
    double mean[WIDE_D];
    double S[WIDE_D * WIDE_D];           // row-major d x d, row stride d
};

// ancestors straight from the global CDF (clouds the bucketed sampler does not take): the draw of k_resample_philox's
// round 0 -- uniform 0 of block (slot, epoch << 16, 0)
__global__ __launch_bounds__(QSMC_BLOCK) void k_anc_direct(const double *__restrict__ cdf, int64_t n_in, int64_t n_out,
                                                           uint32_t k0, uint32_t k1, uint32_t epoch,
                                                           unsigned int *__restrict__ anc) {
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * QSMC_BLOCK) {
        PhiloxStream rng{(uint64_t)i, (epoch << 16), k0, k1};
        double u, unused;
        rng.uniforms(0, u, unused);
        anc[i] = (unsigned int)search_right(cdf, n_in, u);
    }
}

// The kicks of all output slots, 16 per wave trip.  Lane l = (g = l >> 4, n = l & 15) works on slot k0 + n.  K = S Z for
// the 16 slots is NB x NB chains of four v_mfma_f64_16x16x4: for row block rb and column block t, step s multiplies
// A = S[16 rb + (l & 15)][16 t + 4 g + s] (from LDS, stored transposed: lanes of a step read consecutive words) with
// B = Z[16 t + 4 g + s][n] -- so lane (g, n) draws the Box-Muller pairs 8 t + 2 g, 8 t + 2 g + 1 of slot n for every t.
// D: value r of lane l = K[16 rb + g + 4 r][n]: the lane adds the Liu-West centre of those coordinates (a gather from the
// ancestor's rows) and stores them.
// Normals: DIRECT = true (small clouds, ancestors by k_anc_direct) pair p of slot o is block (o, epoch << 16, 1 + p), the
// stream of k_resample_philox; DIRECT = false (ancestors by k_bucket_anc16) pair p of slot o is block (o * 8 NB + p,
// epoch << 16, 2): the bucketed samplers' "normal n = o * stride + q", stride = 16 NB (oracle/philox.py).
constexpr int KICKW_BT = 256, KICKW_WAVES = KICKW_BT / QSMC_WAVE, KICKW_PER_BLOCK = 256;     // (slots per workgroup: 512 -> 586 us at N = 1e6, 2048 -> 661, 256 -> 530, 128 -> 550)
template <int NB, bool DIRECT>
__global__ __launch_bounds__(KICKW_BT, 3) void k_kick_wide(
    const double *__restrict__ x_in, int64_t ldx_in, const unsigned int *__restrict__ anc, int64_t n_out, int d,
    const LWWide *__restrict__ lw, uint32_t k0, uint32_t k1, uint32_t epoch, double *__restrict__ x_out, OutPlace pl) {
    constexpr int DP = 16 * NB, ST = DP + 4;                       // padded d; row stride of the transposed S in LDS
    __shared__ double sST[DP * ST];                                // sST[k * ST + row] = S[row][k]
    __shared__ double sMu[DP];
    const double lw_a = lw->a;
    for (int t = threadIdx.x; t < DP * DP; t += KICKW_BT) {
        const int row = t / DP, k = t % DP;
        sST[k * ST + row] = (row < d && k < d) ? lw->S[row * d + k] : 0.0;
    }
    for (int t = threadIdx.x; t < DP; t += KICKW_BT) sMu[t] = t < d ? (1.0 - lw_a) * lw->mean[t] : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    const int n = lane & 15, g = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * KICKW_PER_BLOCK;
    const int64_t r1 = r0 + KICKW_PER_BLOCK < n_out ? r0 + KICKW_PER_BLOCK : n_out;
    for (int64_t kb = r0 + (int64_t)wave * 16; kb < r1; kb += KICKW_WAVES * 16) {
        const int64_t k = kb + n;
        const int64_t oc = k < r1 ? k : r1 - 1;                     // (idle columns shadow the last slot: no divergence)
        const int64_t j = (int64_t)anc[oc];
        v4d acc[NB];
#pragma unroll
        for (int rb = 0; rb < NB; ++rb) acc[rb] = v4d{0.0, 0.0, 0.0, 0.0};
        // (one column block at a time, not unrolled: unrolled, the eight Box-Muller pairs and 16 NB operands of a trip were
        //  live together -- 290 VGPRs at NB = 4, one wave per SIMD, 945 us at N = 1e6)
#pragma unroll 1
        for (int t = 0; t < NB; ++t) {
            double z[4];
            if (DIRECT) {
                PhiloxStream nrm{(uint64_t)oc, (epoch << 16), k0, k1};
                nrm.normals(1 + 8 * t + 2 * g, z[0], z[1]);
                nrm.normals(1 + 8 * t + 2 * g + 1, z[2], z[3]);
            } else {
                PhiloxStream nrm{(uint64_t)oc * (uint64_t)(8 * NB) + (uint64_t)(8 * t + 2 * g), (epoch << 16), k0, k1};
                nrm.normals(2, z[0], z[1]);
                nrm.particle += 1;
                nrm.normals(2, z[2], z[3]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double *col = sST + (16 * t + 4 * g + s) * ST + n;      // S[16 rb + n][16 t + 4 g + s]
#pragma unroll
                for (int rb = 0; rb < NB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(col[16 * rb], z[s], acc[rb], 0, 0, 0);
            }
        }
        if (k < r1) {
            const int64_t row_o = place_row(pl, k);
#pragma unroll
            for (int rb = 0; rb < NB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * rb + g + 4 * r;
                    if (m < d) {
                        const double xa = x_in[(int64_t)m * ldx_in + j];
                        x_out[(int64_t)m * pl.ld_m + row_o * pl.ld_s] = (lw_a * xa + sMu[m]) + acc[rb][r];
                    }
                }
        }
    }
}

// legacy-RNG pieces (host draws replayed: resamplers.py:318-372), 16 < d <= 64
__global__ __launch_bounds__(QSMC_BLOCK) void k_centres_wide(const double *__restrict__ x_in, int64_t ldx_in, int d,
                                                             const int64_t *__restrict__ js, int64_t n_out,
                                                             const LWWide *__restrict__ lw, double *__restrict__ mus,
                                                             int64_t ld_mus) {
    const double a = lw->a;
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t j = js[i];
        for (int m = 0; m < d; ++m) mus[m * ld_mus + i] = a * x_in[m * ldx_in + j] + (1.0 - a) * lw->mean[m];   // :325
    }
}

__global__ __launch_bounds__(QSMC_BLOCK) void k_perturb_wide(int d, const double *__restrict__ mus, int64_t ld_mus,
                                                             const int64_t *__restrict__ idxs, int64_t k, int centre_by_idx,
                                                             const LWWide *__restrict__ lw, const double *__restrict__ z,
                                                             int64_t ldz, double *__restrict__ x_out, int64_t ldx_out,
                                                             uint8_t *__restrict__ valid) {
    __shared__ double sS[WIDE_D * WIDE_D];
    for (int t = threadIdx.x; t < d * d; t += QSMC_BLOCK) sS[t] = lw->S[t];
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; r < k; r += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t dst = idxs ? idxs[r] : r;
        const int64_t c = centre_by_idx ? dst : r;
        for (int m = 0; m < d; ++m) {
            double s = 0.0;                     // (S @ z)[m, r], summed in column order like np.dot
            for (int q = 0; q < d; ++q) s += sS[m * d + q] * z[q * ldz + r];
            x_out[m * ldx_out + dst] = mus[m * ld_mus + c] + s;
        }
        valid[r] = 1;                           // (tomography: are_models_valid is all-true, tomography/models.py:143-147)
    }
}

// =============================================================================================
// canonicalize, dim 5 .. 8 (tomography/models.py:149-209).  A lane never holds the particle: rho's lower triangle is
// accumulated coefficient by coefficient (x_a from global memory -- coalesced over the lanes --, the basis element from
// scalar loads: its address is uniform), and the re-expansion x_a = Re tr(B_a^H R) is written coefficient by coefficient.
// Pass 1 (every particle): LDL^H pivot test; a positive-definite rho only needs x / (x_0 sqrt dim); the rest is listed.
// Pass 2 (the list): jacobi_clamp<DIM> (the eigenvector form; at dim 8 its iterate and eigenvectors exceed the register
// file and live in scratch -- correct, and the price of the particles that need it), re-expansion, renormalisation.
// =============================================================================================
template <int DIM>
__device__ __forceinline__ void wide_build_lower(const double *__restrict__ basis, const double *__restrict__ x, int64_t ldx,
                                                 int64_t i, double (&Ar)[DIM][DIM], double (&Ai)[DIM][DIM]) {
    constexpr int D = DIM * DIM;
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int c = 0; c < DIM; ++c) { Ar[r][c] = 0.0; Ai[r][c] = 0.0; }
    // eight coefficients fetched ahead of their use: one load in flight at a time made this loop a chain of memory
    // latencies (the particles of a wave are a gather) -- most of the 4.2 ms of the first list kernels
#pragma unroll 1
    for (int a0 = 0; a0 < D; a0 += 8) {
        double pa[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pa[u] = x[(int64_t)(a0 + u < D ? a0 + u : D - 1) * ldx + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (a0 + u < D) {                               // (uniform)
                const double *B = basis + (size_t)2 * (a0 + u) * DIM * DIM;
#pragma unroll
                for (int r = 0; r < DIM; ++r)
#pragma unroll
                    for (int c = 0; c <= r; ++c) {
                        Ar[r][c] += pa[u] * B[2 * (r * DIM + c)];
                        Ai[r][c] += pa[u] * B[2 * (r * DIM + c) + 1];
                    }
            }
        }
    }
}

template <int DIM>
__global__ __launch_bounds__(QSMC_BLOCK) void k_tomo_classify_wide(const double *__restrict__ basis, double *__restrict__ x,
                                                                   int64_t ldx, int64_t n, int allow_subnormalized,
                                                                   unsigned int *__restrict__ list,
                                                                   unsigned int *__restrict__ count) {
    constexpr int D = DIM * DIM;
    // (uniform trip count: the list is appended to a wave at a time -- one atomic per wave, its entries in lane order, so
    //  that a wave of the list pass reads neighbouring particles; single appends left the list in arrival order and every
    //  load of the list pass a 64-line gather)
    for (int64_t i0 = (int64_t)blockIdx.x * QSMC_BLOCK; i0 < n; i0 += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t i = i0 + threadIdx.x;
        const bool live = i < n;
        bool ok = true;
        if (live) {
        double Ar[DIM][DIM], Ai[DIM][DIM];
        wide_build_lower<DIM>(basis, x, ldx, i, Ar, Ai);
        {
#pragma clang fp contract(on)                             // (a verdict, not a reproduced value: as tomo_clearly_positive)
#pragma unroll
            for (int j = 0; j < DIM; ++j) {
                double dj = Ar[j][j];
#pragma unroll
                for (int k = 0; k < j; ++k) dj -= (Ar[j][k] * Ar[j][k] + Ai[j][k] * Ai[j][k]) * Ar[k][k];
                ok = ok && (dj > 0.0);
                Ar[j][j] = dj;
                const double inv = 1.0 / dj;
#pragma unroll
                for (int r = j + 1; r < DIM; ++r) {
                    double sr = Ar[r][j], si = Ai[r][j];
#pragma unroll
                    for (int k = 0; k < j; ++k) {
                        const double tr = Ar[r][k] * Ar[j][k] + Ai[r][k] * Ai[j][k];
                        const double ti = Ai[r][k] * Ar[j][k] - Ar[r][k] * Ai[j][k];
                        sr -= tr * Ar[k][k];
                        si -= ti * Ar[k][k];
                    }
                    Ar[r][j] = sr * inv;
                    Ai[r][j] = si * inv;
                }
            }
        }
        if (ok && !allow_subnormalized) {                 // tomography/models.py:194-209
            const double inv = 1.0 / (x[i] * sqrt((double)DIM));
            for (int a = 0; a < D; ++a) x[(int64_t)a * ldx + i] = x[(int64_t)a * ldx + i] * inv;
        }
        }
        const bool listed = live && !ok;
        const unsigned long long mk = __ballot(listed);
        if (mk) {
            const int lane = threadIdx.x & (QSMC_WAVE - 1);
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(count, (unsigned int)__popcll(mk));
            base = __shfl(base, 0, QSMC_WAVE);
            if (listed) list[base + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned int)i;
        }
    }
}

template <int DIM>
__global__ __launch_bounds__(64) void k_tomo_canon_list_wide(const double *__restrict__ basis, double *__restrict__ x,
                                                             int64_t ldx, int allow_subnormalized,
                                                             const unsigned int *__restrict__ list,
                                                             const unsigned int *__restrict__ count) {
    constexpr int D = DIM * DIM;
    const unsigned int m = *count;
    for (unsigned int t = blockIdx.x * 64u + threadIdx.x; t < m; t += gridDim.x * 64u) {
        const int64_t i = (int64_t)list[t];
        double Ar[DIM][DIM], Ai[DIM][DIM], Rr[DIM][DIM], Ri[DIM][DIM];
        wide_build_lower<DIM>(basis, x, ldx, i, Ar, Ai);
        const bool any_neg = jacobi_clamp<DIM>(Ar, Ai, Rr, Ri);
        if (any_neg) {
            // x_a = Re sum_rc conj(B_a[r][c]) R[r][c]; a = 0 first: the trace renormalisation divides by ITS new value
            double inv = 1.0;
            for (int a = 0; a < D; ++a) {
                const double *B = basis + (size_t)2 * a * DIM * DIM;
                double s = 0.0;
#pragma unroll
                for (int r = 0; r < DIM; ++r)
#pragma unroll
                    for (int c = 0; c < DIM; ++c) s += B[2 * (r * DIM + c)] * Rr[r][c] + B[2 * (r * DIM + c) + 1] * Ri[r][c];
                if (a == 0 && !allow_subnormalized) inv = 1.0 / (s * sqrt((double)DIM));
                x[(int64_t)a * ldx + i] = allow_subnormalized ? s : s * inv;
            }
        } else if (!allow_subnormalized) {
            const double inv = 1.0 / (x[i] * sqrt((double)DIM));
            for (int a = 0; a < D; ++a) x[(int64_t)a * ldx + i] = x[(int64_t)a * ldx + i] * inv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// canonicalize, dim 5 .. 8, the default form: FOUR kernels around a scratch copy of rho in packed Hermitian form.
//
// The first forms of this file had every lane build rho = sum_a x_a B_a itself -- DIM^2 coefficients times DIM (DIM + 1) / 2
// complex entries, the basis through scalar loads -- and re-expand x'_a = Re tr(B_a^H R) the same way.  Measured with the
// Jacobi sweeps switched off (profiles/r6_i_canon_list_sweep_cap_experiment.txt): 3.15 of the list kernel's 4.4 ms at
// N = 1e6 were those two loops -- one wave per SIMD waiting on scalar loads of a 64 KB basis that does not fit the 16 KB scalar
// cache -- and the sweeps themselves 1.2 ms.  But rho in packed form (DIM real diagonal entries, then (re, im) of the
// strict lower triangle: E = DIM^2 reals) is a REAL LINEAR MAP of x: rho_packed = Mb x with Mb[e][a] = Re / Im B_a[r][c],
// and the re-expansion its transpose with the off-diagonal rows doubled.  For all particles at once these are two
// (E x E) (E x N) products -- the shape of the Liu-West kick S Z -- and run on the f64 matrix cores like it:
//   k_canon_mats<DIM>      Mb and Me (padded to 16 NB) from the basis tensor, on the device, per call (4096 entries);
//   k_gemm_wide<NB, 0>     rho_packed[e][i] = sum_a Mb[e][a] x[a][i] for every particle -> scratch (E x N doubles);
//   k_tomo_ldl_wide<DIM>   LDL^H pivot test on the packed rho (64 coalesced loads per lane): positive definite -> x / (x_0 sqrt dim),
//                          else listed (a wave at a time, in lane order);
//   k_tomo_jacobi_wide<DIM>  the listed: ONE-SIDED Jacobi, no eigenvectors (below), R = (A + |A|) / 2 written back packed;
//   k_gemm_wide<NB, 1>     x'[a][i] = sum_e Me[a][e] R_packed[e][i] for the listed, trace renormalisation in its epilogue.
//
// The clamp of tomography/models.py:185-192, V max(Lambda, 0) V^H, is (A + |A|) / 2 with |A| = (A A^H)^(1/2) the Hermitian
// polar factor -- and |A| = G Sigma^-1 G^H for G = A W with orthogonal columns of norms Sigma: Hestenes' Jacobi on the COLUMNS
// of A.  Only G is iterated (256 VGPRs at dim 8, against iterate + eigenvectors + reconstruction of the eigenvector form:
// 510 and AGPR traffic on every access); the Gram entry of a pivot is a sum down two columns; the column norms follow a
// rotation by alpha' = alpha - t |gamma|, beta' = beta + t |gamma| (recomputed exactly at every sweep).  The 28 pivots of a sweep are
// unrolled (register indices must be compile-time); as seven rounds of the fixed position pairs (0,7), (1,6), (2,5), (3,4) with a
// cyclic move of columns 1 .. 7 in between -- a loop body of four pivots, |A| does not care where a column sits -- the pass takes
// 2.18 ms against 1.96: the moves cost more than the instruction cache.  The kernel issues ~11 000 instructions per particle and
// sweep, a sixth of them v_accvgpr moves: an 8 x 8 complex iterate IS the 256 architectural VGPRs, everything else lives in
// AGPRs.  Convergence is to ABSOLUTE accuracy -- a pivot is rotated
// while |gamma|^2 > 1e-30 ||A||_F^2 max(alpha, beta): the error of |A| from a residual gamma is |gamma| / (sigma_p + sigma_q) --
// because the clouds this runs on sit ON the boundary of the cone: a zero eigenvalue leaves a column of rounding noise
// whose RELATIVE orthogonality never converges (measured: every particle ran to the sweep limit).  Degenerate |lambda|
// pairs of opposite sign mix in G's columns -- |A| restricted to that subspace is |lambda| times the identity, so the sum
// over the pair is right whatever the mixture.  A particle whose tr |A| - tr A vanishes to rounding has no negative
// eigenvalue: left as it is but for the trace (its list entry gets bit 31).
// (Also built and measured on the way: four lanes per particle with DPP Gram sums -- texture-addresser-bound on the basis
//  entries its lanes need, 19.7 ms, removed)
// ---------------------------------------------------------------------------------------------
template <int DIM> __host__ __device__ constexpr int pk_re(int r, int c) { return DIM + 2 * (r * (r - 1) / 2 + c); }   // r > c

template <int DIM>
__global__ __launch_bounds__(QSMC_BLOCK) void k_canon_mats(const double *__restrict__ basis, double *__restrict__ Mb,
                                                           double *__restrict__ Me) {
    constexpr int E = DIM * DIM, DP = 16 * ((E + 15) / 16);
    for (int t = blockIdx.x * QSMC_BLOCK + threadIdx.x; t < DP * DP; t += gridDim.x * QSMC_BLOCK) {
        const int e = t / DP, a = t % DP;
        double v = 0.0, mult = 1.0;
        if (e < E && a < E) {
            int r = e, c = e, part = 0;
            if (e >= DIM) {
                const int q = e - DIM, p = q >> 1;
                part = q & 1;
                r = 1;
                while (r * (r + 1) / 2 <= p) ++r;
                c = p - r * (r - 1) / 2;
                mult = 2.0;
            }
            v = basis[2 * ((a * DIM + r) * DIM + c) + part];
        }
        Mb[e * DP + a] = v;                                   // rho_packed = Mb x
        Me[a * DP + e] = mult * v;                            // x = Me R_packed (Re tr(B_a^H R): off-diagonal entries twice)
    }
}

// out = M in for 16 particles per wave trip (the kick kernel's MFMA chains with M in the place of S).
// MODE 0: in = the cloud x (every particle i < n), out = scratch[e * ld + i].
// MODE 1: in = scratch (the listed particles: list[t] & 0x7fffffff; bit 31: nothing was clamped), out = x with the trace
//         renormalisation x / (x_0 sqrt dim) unless allow_subnormalized (tomography/models.py:194-209); a particle with
//         bit 31 keeps its own coefficients (the reference returns it untouched) and is only renormalised.
constexpr int GEMMW_BT = 256, GEMMW_WAVES = GEMMW_BT / QSMC_WAVE, GEMMW_PER_BLOCK = 256;
template <int NB, int MODE>
__global__ __launch_bounds__(GEMMW_BT, 3) void k_gemm_wide(const double *__restrict__ M, int E, double *__restrict__ x,
                                                           int64_t ldx, int64_t n, double *__restrict__ scratch, int64_t ld,
                                                           const unsigned int *__restrict__ list,
                                                           const unsigned int *__restrict__ count, int dim,
                                                           int allow_subnormalized) {
    constexpr int DP = 16 * NB, ST = DP + 4;
    __shared__ double sMT[DP * ST];                                // sMT[k * ST + row] = M[row][k]
    const int64_t n_items = MODE == 0 ? n : (int64_t)*count;
    const int64_t r0 = (int64_t)blockIdx.x * GEMMW_PER_BLOCK;
    if (r0 >= n_items) return;
    for (int t = threadIdx.x; t < DP * DP; t += GEMMW_BT) {
        const int row = t / DP, k = t % DP;
        sMT[k * ST + row] = M[row * DP + k];
    }
    __syncthreads();
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    const int nn = lane & 15, g = lane >> 4;
    const int64_t r1 = r0 + GEMMW_PER_BLOCK < n_items ? r0 + GEMMW_PER_BLOCK : n_items;
    for (int64_t kb = r0 + (int64_t)wave * 16; kb < r1; kb += GEMMW_WAVES * 16) {
        const int64_t t = kb + nn;
        const int64_t tc = t < r1 ? t : r1 - 1;                     // (idle columns shadow the last item: no divergence)
        unsigned int entry = 0u;
        if (MODE == 1) entry = list[tc];
        const int64_t i = MODE == 0 ? tc : (int64_t)(entry & 0x7fffffffu);
        v4d acc[NB];
#pragma unroll
        for (int rb = 0; rb < NB; ++rb) acc[rb] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll 1                                                 // (unrolled: 168 VGPRs + 300 B of scratch at NB = 4)
        for (int tt = 0; tt < NB; ++tt) {
            double z[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 16 * tt + 4 * g + s;
                z[s] = k < E ? (MODE == 0 ? x[(int64_t)k * ldx + i] : scratch[(int64_t)k * ld + i]) : 0.0;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double *col = sMT + (16 * tt + 4 * g + s) * ST + nn;    // M[16 rb + nn][16 tt + 4 g + s]
#pragma unroll
                for (int rb = 0; rb < NB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(col[16 * rb], z[s], acc[rb], 0, 0, 0);
            }
        }
        // D: value r of lane (g, nn) = out[16 rb + g + 4 r][item nn]
        if (MODE == 0) {
            if (t < r1) {
#pragma unroll
                for (int rb = 0; rb < NB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = 16 * rb + g + 4 * r;
                        if (e < E) scratch[(int64_t)e * ld + i] = acc[rb][r];
                    }
            }
        } else {
            const bool clamped = (entry >> 31) == 0u;
            // coefficient 0 of the item: value 0 of block 0 on lane (g = 0, nn)
            const double k0 = __shfl(acc[0][0], nn, QSMC_WAVE);
            const double x0 = clamped ? k0 : x[i];
            const double inv = allow_subnormalized ? 1.0 : 1.0 / (x0 * sqrt((double)dim));
            if (t < r1 && (clamped || !allow_subnormalized)) {
#pragma unroll
                for (int rb = 0; rb < NB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a = 16 * rb + g + 4 * r;
                        if (a < E) {
                            double *dst = x + (int64_t)a * ldx + i;
                            *dst = (clamped ? acc[rb][r] : *dst) * inv;
                        }
                    }
            }
        }
    }
}

// LDL^H pivot test on the packed rho of every particle: positive definite -> only the trace renormalisation, else listed
template <int DIM>
__global__ __launch_bounds__(QSMC_BLOCK) void k_tomo_ldl_wide(const double *__restrict__ scratch, int64_t ld, double *__restrict__ x,
                                                              int64_t ldx, int64_t n, int allow_subnormalized,
                                                              unsigned int *__restrict__ list, unsigned int *__restrict__ count) {
    constexpr int D = DIM * DIM;
    for (int64_t i0 = (int64_t)blockIdx.x * QSMC_BLOCK; i0 < n; i0 += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t i = i0 + threadIdx.x;
        const bool live = i < n;
        bool ok = true;
        if (live) {
#pragma clang fp contract(on)                             // (a verdict, not a reproduced value: as tomo_clearly_positive)
            double Ar[DIM][DIM], Ai[DIM][DIM];
#pragma unroll
            for (int r = 0; r < DIM; ++r) {
                Ar[r][r] = scratch[(int64_t)r * ld + i];
#pragma unroll
                for (int c = 0; c < r; ++c) {
                    Ar[r][c] = scratch[(int64_t)pk_re<DIM>(r, c) * ld + i];
                    Ai[r][c] = scratch[(int64_t)(pk_re<DIM>(r, c) + 1) * ld + i];
                }
            }
#pragma unroll
            for (int j = 0; j < DIM; ++j) {
                double dj = Ar[j][j];
#pragma unroll
                for (int k = 0; k < j; ++k) dj -= (Ar[j][k] * Ar[j][k] + Ai[j][k] * Ai[j][k]) * Ar[k][k];
                ok = ok && (dj > 0.0);
                Ar[j][j] = dj;
                const double inv = 1.0 / dj;
#pragma unroll
                for (int r = j + 1; r < DIM; ++r) {
                    double sr = Ar[r][j], si = Ai[r][j];
#pragma unroll
                    for (int k = 0; k < j; ++k) {
                        const double tr = Ar[r][k] * Ar[j][k] + Ai[r][k] * Ai[j][k];
                        const double ti = Ai[r][k] * Ar[j][k] - Ar[r][k] * Ai[j][k];
                        sr -= tr * Ar[k][k];
                        si -= ti * Ar[k][k];
                    }
                    Ar[r][j] = sr * inv;
                    Ai[r][j] = si * inv;
                }
            }
            if (ok && !allow_subnormalized) {             // tomography/models.py:194-209
                const double inv = 1.0 / (x[i] * sqrt((double)DIM));
                for (int a = 0; a < D; ++a) x[(int64_t)a * ldx + i] = x[(int64_t)a * ldx + i] * inv;
            }
        }
        const bool listed = live && !ok;
        const unsigned long long mk = __ballot(listed);
        if (mk) {                                         // one atomic per wave, entries in lane order
            const int lane = threadIdx.x & (QSMC_WAVE - 1);
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(count, (unsigned int)__popcll(mk));
            base = __shfl(base, 0, QSMC_WAVE);
            if (listed) list[base + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned int)i;
        }
    }
}

template <int DIM>
__global__ __launch_bounds__(64) void k_tomo_jacobi_wide(double *__restrict__ scratch, int64_t ld, unsigned int *__restrict__ list,
                                                         const unsigned int *__restrict__ count) {
#pragma clang fp contract(on)                                 // (as in jacobi_clamp: nothing reproduces these intermediates)
    constexpr int NC = 8;                                     // columns padded to 8 (zero columns are never rotated)
    const unsigned int m = *count;
    for (unsigned int t = blockIdx.x * 64u + threadIdx.x; t < m; t += gridDim.x * 64u) {
        const int64_t i = (int64_t)list[t];
        double Gr[DIM][NC], Gi[DIM][NC];
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
            for (int c = DIM; c < NC; ++c) { Gr[r][c] = 0.0; Gi[r][c] = 0.0; }
        double tr_a = 0.0, frob2 = 0.0;
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
            Gr[r][r] = scratch[(int64_t)r * ld + i];
            Gi[r][r] = 0.0;
            tr_a += Gr[r][r];
            frob2 += Gr[r][r] * Gr[r][r];
#pragma unroll
            for (int c = 0; c < r; ++c) {
                const double re = scratch[(int64_t)pk_re<DIM>(r, c) * ld + i], im = scratch[(int64_t)(pk_re<DIM>(r, c) + 1) * ld + i];
                Gr[r][c] = re;
                Gi[r][c] = im;
                Gr[c][r] = re;
                Gi[c][r] = -im;
                frob2 += 2.0 * (re * re + im * im);
            }
        }
        const double tiny2 = 1e-28 * frob2, conv2 = 1e-30 * frob2;
        double nrm[NC];
        for (int sweep = 0; sweep < 30; ++sweep) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double s2 = 0.0;
#pragma unroll
                for (int r = 0; r < DIM; ++r) s2 += Gr[r][c] * Gr[r][c] + Gi[r][c] * Gi[r][c];
                nrm[c] = s2;
            }
            bool rotated = false;
#pragma unroll
            for (int p = 0; p < NC; ++p)
#pragma unroll
                for (int q = p + 1; q < NC; ++q) {
                    if (p >= DIM || q >= DIM) continue;       // (padding columns: compile-time)
                    const double big = fmax(nrm[p], nrm[q]);
                    if (big > tiny2) {
                        double gr = 0.0, gi = 0.0;            // g_p^H g_q
#pragma unroll
                        for (int r = 0; r < DIM; ++r) {
                            gr += Gr[r][p] * Gr[r][q] + Gi[r][p] * Gi[r][q];
                            gi += Gr[r][p] * Gi[r][q] - Gi[r][p] * Gr[r][q];
                        }
                        const double mag2 = gr * gr + gi * gi;
                        if (mag2 > conv2 * big && mag2 > 1e-290) {
                            const double imag = j_rsqrt(mag2);
                            const double er = gr * imag, ei = gi * imag;      // e^{i phi} = gamma / |gamma|
                            const double zeta = (nrm[q] - nrm[p]) * (0.5 * imag);
                            const double z2 = 1.0 + zeta * zeta;
                            const double rt = z2 * j_rsqrt(z2);
                            const double tt = (zeta >= 0.0 ? 1.0 : -1.0) * j_rcp(fabs(zeta) + rt);
                            const double cs = j_rsqrt(1.0 + tt * tt);
                            const double sn = tt * cs;
                            const double th = tt * (mag2 * imag);             // t |gamma|
#pragma unroll
                            for (int r = 0; r < DIM; ++r) {
                                const double pr = Gr[r][p], pi = Gi[r][p], qr = Gr[r][q], qi = Gi[r][q];
                                Gr[r][p] = cs * pr - sn * (er * qr + ei * qi);        // g_p' = c g_p - s e^{-i phi} g_q
                                Gi[r][p] = cs * pi - sn * (er * qi - ei * qr);
                                Gr[r][q] = sn * (er * pr - ei * pi) + cs * qr;        // g_q' = s e^{i phi} g_p + c g_q
                                Gi[r][q] = sn * (er * pi + ei * pr) + cs * qi;
                            }
                            nrm[p] -= th;
                            nrm[q] += th;
                            rotated = true;
                        }
                    }
                }
            if (!rotated) break;
        }
        // column norms = |lambda|; H = G Sigma^(-1/2), so that |A| = H H^H
        double sum_sigma = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double s2 = 0.0;
#pragma unroll
            for (int r = 0; r < DIM; ++r) s2 += Gr[r][c] * Gr[r][c] + Gi[r][c] * Gi[r][c];
            const double sg = sqrt(s2);
            sum_sigma += sg;
            const double isg = sg > 1e-300 ? 1.0 / sqrt(sg) : 0.0;
#pragma unroll
            for (int r = 0; r < DIM; ++r) { Gr[r][c] *= isg; Gi[r][c] *= isg; }
        }
        const bool neg = (sum_sigma - tr_a) > 64.0 * 2.220446049250313e-16 * sum_sigma;
        if (!neg) {
            list[t] = (unsigned int)i | 0x80000000u;          // nothing to clamp: the expand pass only renormalises
            continue;
        }
        // R = (A + H H^H) / 2, packed, entry by entry over A's own scratch.  (The particle index goes through an empty asm: left
        // visible, the 64 addresses of this write-back are formed BEFORE the sweeps and kept live through them -- 476 VGPRs
        // instead of 352 at dim 8, 226 instead of 160 at dim 5 (two waves per SIMD there); no difference in time at dim 8)
        long long i2 = (long long)i;
        asm volatile("" : "+v"(i2));
        double *const sc2 = scratch + i2;
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double sr = 0.0, si = 0.0;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    sr += Gr[r][k] * Gr[c][k] + Gi[r][k] * Gi[c][k];
                    si += Gi[r][k] * Gr[c][k] - Gr[r][k] * Gi[c][k];
                }
                if (r == c) {
                    double *dst = sc2 + (int64_t)r * ld;
                    *dst = 0.5 * (*dst + sr);
                } else {
                    double *dre = sc2 + (int64_t)pk_re<DIM>(r, c) * ld, *dim_ = sc2 + (int64_t)(pk_re<DIM>(r, c) + 1) * ld;
                    *dre = 0.5 * (*dre + sr);
                    *dim_ = 0.5 * (*dim_ + si);
                }
            }
    }
}
