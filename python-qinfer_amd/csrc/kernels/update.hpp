// kernels/update.hpp -- the fused Bayes update (k_update_fused, k_update_multi, hypothetical sums) and the two-level deterministic reduction they share.
// Part of the single translation unit qsmc_kernels.hip (included there, in this order; not a stand-alone header).
#pragma once

// =============================================================================================
// fused Bayes update
// =============================================================================================
constexpr int UPD_UNROLL = 4;

// ---------------------------------------------------------------------------------------------
// Two-level deterministic reduction.  Each workgroup writes NS sums + 1 min (block_publish); a
// one-workgroup kernel (k_reduce_partials) sums the per-workgroup partials in INDEX order and writes
// the totals to device memory AND straight into pinned host memory (no D2H copy command).
// out layout: [sum_0 .. sum_{NS-1}, min].
//
// Measured alternative (round-1 profile c): doing the second level inside the same launch with an
// agent-scope arrival ticket cost +11 us on the 44 us update kernel -- 2048 workgroups finishing
// together saturate one atomic word (~88 arrivals/us) -- versus 4.9 us + one launch boundary here.
// ---------------------------------------------------------------------------------------------
struct ReduceOut {
    double *partials;        // [NS + 1][grid]
    double *out_dev;         // [NS + 1] device (always written)
    double *out_mapped;      // [NS + 1] device alias of pinned host memory (nullable)
    double *stats4;          // optional caller buffer in qsmc_update_stats_t order (nullable)
    unsigned long long *flag;  // device alias of the pinned completion word (nullable)
    unsigned long long seq;    // value to publish there once out_mapped is complete
    const unsigned long long *failed_src;   // the resampler's failed-particle counter (device) ...
    double *failed_dst;                     // ... copied to its pinned slot by every host-visible reduction
    double *tile_sums;                      // k_update_fused only: sum of w' per TILE particles (nullable)
    int *prefix_gate;                       // k_update_fused only: device word for the resample-prefix gate (nullable) ...
    double prefix_thresh;                   // ... opened when (sum w')^2 / sum w'^2 < prefix_thresh and no guard is due
    double *tile_prefix;                    // k_update_fused only (nullable): [tp_chunks + 1] monotone prefix of the UNNORMALISED
    int tp_chunks, tp_tpc;                  //   chunk sums, formed from tile_sums by a second workgroup of the reducing launch
    long long tp_ntiles;                    //   (k_reduce_partials_scan) while the first one reduces
};

// |sum w'| below this and the host renormalises by 1 instead (smc.py:369-370): no speculative prefix then
constexpr double PREFIX_NORM_EPS = 2.220446049250313e-16;

template <int NS>
__device__ __forceinline__ void block_publish(double (&v)[NS], double mn, const ReduceOut &ro) {
    // one barrier: every wave reduces its NS sums and the minimum, lane 0 parks them in LDS, then thread k
    // combines value k over the waves (in wave order, as before: bitwise the same totals) and stores it --
    // NS + 1 parallel stores instead of one thread doing them in sequence behind three more barriers
    __shared__ double lds[QSMC_WAVES_PER_BLOCK * (NS + 1)];
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = wave_sum(v[k]);
    mn = wave_min(mn);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) lds[wave * (NS + 1) + k] = v[k];
        lds[wave * (NS + 1) + NS] = mn;
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= NS; k += QSMC_BLOCK) {
        double s = lds[k];
#pragma unroll
        for (int wv = 1; wv < QSMC_WAVES_PER_BLOCK; ++wv) {
            const double t = lds[wv * (NS + 1) + k];
            s = (k < NS) ? s + t : fmin(s, t);
        }
        ro.partials[(size_t)k * gridDim.x + blockIdx.x] = s;          // column k of [NS + 1][grid]: the reducer reads it coalesced
    }
}

// One workgroup on the critical path of every datum: 4.8 us at N = 1e7 (2442 partial rows of 6).  Split with
// wall_clock64(): the sweep 1.8-2.1 us WHATEVER the unrolling (4 or 10 rows in flight), the thread count (128 ... 1024;
// 1024 was slower overall, 6.6 us) or the layout (row-major at a 48-byte stride, or column-major and coalesced as now) --
// one cold first touch of lines other XCDs wrote, not bandwidth and not rounds of latency; wave/LDS reduction and the
// stores 1.1 us (1.4 before the resampler's two counters were fetched ahead of the sweep); the system-scope fence before
// the completion word 0.9 us (a bare s_waitcnt costs the same: it is the wait for the pinned-memory stores, not a cache
// write-back); the rest is launch.  Doing this level inside the update kernel behind arrival tickets was measured in
// round 1 (+11 us), re-costed in round 2 (every dependent global round trip is 1-2 us and that chain has more of them) and
// built again in round 4 with two-level tickets (tools/experiments/r4_reduction_folded_into_update.patch): with agent-scope
// release fences every workgroup writes back its XCD's L2 (update kernel 39 -> 120 us); with the partials stored through the
// caches and order-only fences the kernel grows by 5.4 us (ticket chain, a cold sweep of the rows, the publish -- serial at
// its end) against 7-8 us of launch saved, and with the chunk prefix back in k_bucket_counts the step comes out 2-3 % slower.
template <int NS>
__device__ __forceinline__ void reduce_partials_body(int nblocks, const ReduceOut &ro) {
    constexpr int THREADS = QSMC_BLOCK, WAVES = THREADS / QSMC_WAVE, UNROLL = NS <= 17 ? 4 : (NS <= 38 ? 2 : 1);
    __shared__ double lds[WAVES * (NS + 1)];
    __shared__ double tot[NS + 1];
    unsigned long long failed0 = 0ull, failed1 = 0ull;
    if (threadIdx.x == 0 && ro.failed_dst) {
        failed0 = ro.failed_src[0];
        failed1 = ro.failed_src[1];
    }
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    double m2 = INFINITY;
    // UNROLL rows per thread and trip, every load of a trip issued before the first add (rows past the end are
    // clamped to the last row and their values dropped: a guarded loop with a run-time trip count ends in a
    // remainder loop that takes the rows one latency at a time)
    for (int g0 = threadIdx.x; g0 < nblocks; g0 += THREADS * UNROLL) {
        double v[UNROLL][NS + 1];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int g = g0 + u * THREADS;
            const double *p = ro.partials + (g < nblocks ? g : nblocks - 1);
#pragma unroll
            for (int k = 0; k <= NS; ++k) v[u][k] = p[(size_t)k * nblocks];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const bool in = g0 + u * THREADS < nblocks;
#pragma unroll
            for (int k = 0; k < NS; ++k) acc[k] += in ? v[u][k] : 0.0;
            m2 = fmin(m2, in ? v[u][NS] : INFINITY);
        }
    }
    {   // waves, then thread k combines value k over the waves in wave order, then thread 0 publishes
        const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = wave_sum(acc[k]);
        m2 = wave_min(m2);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) lds[wave * (NS + 1) + k] = acc[k];
            lds[wave * (NS + 1) + NS] = m2;
        }
        __syncthreads();
        if (threadIdx.x <= NS) {
            const int k = threadIdx.x;
            double t = lds[k];
#pragma unroll
            for (int wv = 1; wv < WAVES; ++wv) {
                const double u = lds[wv * (NS + 1) + k];
                t = (k < NS) ? t + u : fmin(t, u);
            }
            tot[k] = t;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) acc[k] = tot[k];
            m2 = tot[NS];
        }
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            ro.out_dev[k] = acc[k];
            if (ro.out_mapped) ro.out_mapped[k] = acc[k];
        }
        ro.out_dev[NS] = m2;
        if (ro.out_mapped) ro.out_mapped[NS] = m2;
        if (ro.stats4) {                 // caller layout: qsmc_update_stats_t order, then the extra sums
            ro.stats4[0] = acc[0];
            ro.stats4[1] = acc[1];
            ro.stats4[2] = m2;
            ro.stats4[3] = acc[2];
#pragma unroll
            for (int k = 3; k < NS; ++k) ro.stats4[4 + (k - 3)] = acc[k];
        }
        // a resample queued before this reduction has finished by now (stream order): its count of particles
        // that stayed invalid rides along (qsmc_last_resample_failed reads it after this synchronisation)
        if (ro.failed_dst) {
            ro.failed_dst[0] = (double)failed0;
            ro.failed_dst[-1] = (double)failed1;                // how many outputs of that resample needed a global redraw
        }
        if (ro.prefix_gate) {
            // The host's resample test (smc.py:263-277 via n_ess = 1 / sum w^2 of the normalised weights), taken here
            // with the same three IEEE operations on the same two sums, so that the weight-only prefix of the
            // resampler (k_bucket_counts, queued right behind this kernel) starts NOW instead of a host round trip
            // later.  The gate and the normaliser it used are published: the host takes the prefix as done only if
            // both agree with its own decision and numbers, else it queues the prefix itself as before.
            const double ess = acc[0] * acc[0] / acc[1];
            const int open = (acc[2] == 0.0 && fabs(acc[0]) >= PREFIX_NORM_EPS && ess < ro.prefix_thresh) ? 1 : 0;
            *ro.prefix_gate = open;
            if (ro.failed_dst) {
                ro.failed_dst[-2] = (double)open;
                ro.failed_dst[-3] = acc[0];
            }
        }
        if (ro.flag) {                   // the host spins on this word instead of hipStreamSynchronize
            __threadfence_system();
            *reinterpret_cast<volatile unsigned long long *>(ro.flag) = ro.seq;
        }
    }
}

template <int NS>
__global__ __launch_bounds__(QSMC_BLOCK) void k_reduce_partials(int nblocks, ReduceOut ro) {
    reduce_partials_body<NS>(nblocks, ro);
}

// Per-particle accumulation of the update: [sum w', sum w'^2, #bad, sum w' x_m (DMOM),
// sum w' x_m x_q (m <= q)] and min w'.  DMOM > 0 folds the weighted moments of the NEW weights
// into the same pass (x is already in registers): est_mean / est_covariance_mtx and the
// resampler's moments then cost no extra sweep over HBM.
template <int DMOM>
struct UpdAcc {
    static constexpr int NS = 3 + DMOM + DMOM * (DMOM + 1) / 2;
    double s[NS];
    double mn;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = 0.0;
        mn = INFINITY;
    }
    __device__ __forceinline__ void add(double w, const double *p) {
        s[0] += w;
        s[1] += w * w;
        s[2] += (w >= 0.0) ? 0.0 : 1.0;       // counts NaN too, like np.all(w >= 0)
        mn = fmin(mn, w);                      // fmin drops NaN; s[2] records it
        int k = 3 + DMOM;
#pragma unroll
        for (int m = 0; m < DMOM; ++m) {
            const double wx = w * p[m];
            s[3 + m] += wx;
#pragma unroll
            for (int q = m; q < DMOM; ++q) s[k++] += wx * p[q];
        }
    }
};

template <int KIND, int VEC, bool ONES, bool POW>   // ONES: w_in == nullptr stands for all-ones weights; POW: MLEModel
__global__ __launch_bounds__(QSMC_BLOCK) void k_update_fused(
    const double *__restrict__ x, int64_t ldx, int64_t n, const double *__restrict__ w_in,
    double *__restrict__ w_out, double prev_norm, ExpArgs e, int64_t outcome, ReduceOut ro, int nt) {
    constexpr int D = Model<KIND>::D;
    constexpr int DMOM = D <= 4 ? D : 0;           // moments ride along for d <= 4
    const int d = (KIND == QSMC_MODEL_TOMOGRAPHY) ? e.d : D;
    constexpr int64_t TILE = (int64_t)QSMC_BLOCK * VEC * UPD_UNROLL;
    UpdAcc<DMOM> acc;
    acc.init();
    // w / norm as w * (1 / norm): an fp64 division is ~25 VALU instructions per particle in a kernel whose
    // VALU time matters (see cos_sq); the two differ by at most one rounding of the stored weight
    const double inv_norm = 1.0 / prev_norm;
    // Sum of the new weights per tile, for the resampler: its chunk sums (k_chunk_sums, an 80 MB read) are
    // sums of two such tiles, so a resample that follows this update starts from them instead of reading
    // the weights once more.  One wave reduction per wave and tile, no barrier: each wave stores its own part
    // (k_scan_sums adds the parts in a fixed order); off when ro.tile_sums is null.
    for (int64_t base = (int64_t)blockIdx.x * TILE; base < n; base += (int64_t)gridDim.x * TILE) {
        double tsum = 0.0;
        // (D <= 2 only.  Round 3 tried the same form for RB, d = 3 / 4: 16 loads in flight per wave but 174 / 204 VGPRs,
        //  two waves per SIMD instead of four -- 94 -> 98 us at N = 1.25e7, Binomial(RB) 125 -> 143 us.  Round 4: HALF a
        //  tile's loads at a time -- still 173 / 203 VGPRs: 96-98 -> 98 us, Binomial(RB) 125 -> 137 us.  And one sub-tile
        //  AHEAD of the arithmetic, held to 168 VGPRs for three waves per SIMD (12 spills): 95.8 -> 94.9 us -- not kept.)
        if (VEC == 2 && D <= 2 && base + TILE <= n) {
            // full tile: every load of the tile is issued before the first likelihood is evaluated, so a wave
            // has UPD_UNROLL x (1 + d) 16-byte loads in flight instead of 1 + d (the guarded path below
            // serialises load -> compute -> store per sub-tile because of its bounds branches)
            double2 wi[UPD_UNROLL], xv[UPD_UNROLL][D];
#pragma unroll
            for (int u = 0; u < UPD_UNROLL; ++u) {
                const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2;
                typedef double nt2 __attribute__((ext_vector_type(2)));
                if (nt) {               // (uniform) a cloud beyond the Infinity Cache: streaming hints on loads and stores
                    if (ONES) { wi[u].x = 1.0; wi[u].y = 1.0; } else { const nt2 t = __builtin_nontemporal_load(reinterpret_cast<const nt2 *>(w_in + i)); wi[u].x = t.x; wi[u].y = t.y; }
#pragma unroll
                    for (int m = 0; m < D; ++m) { const nt2 t = __builtin_nontemporal_load(reinterpret_cast<const nt2 *>(x + m * ldx + i)); xv[u][m].x = t.x; xv[u][m].y = t.y; }
                } else {
                    if (ONES) { wi[u].x = 1.0; wi[u].y = 1.0; } else wi[u] = *reinterpret_cast<const double2 *>(w_in + i);
#pragma unroll
                    for (int m = 0; m < D; ++m) xv[u][m] = *reinterpret_cast<const double2 *>(x + m * ldx + i);
                }
            }
            // Binomial(SimplePrecession), n_meas <= 64 (round 4): the two integer powers of the pmf for the tile's eight
            // particles in ONE square-and-multiply loop -- eight independent chains inside each bit step instead of sixteen
            // dependent loops one after the other (binom_pmf per particle: the update kernel of config 3 sat at 0.53 of the
            // roofline with its VALUs a third busy).  Same multiplications per particle in the same order: same bits.
            double Lb[2 * UPD_UNROLL];
            bool batched = false;
            if constexpr (KIND == QSMC_MODEL_BINOMIAL_PRECESSION && !POW) {
                if (e.n_meas <= 64.0 && outcome >= 0 && (double)outcome <= e.n_meas) {       // (uniform)
                    double pr[2 * UPD_UNROLL], qr[2 * UPD_UNROLL], pk[2 * UPD_UNROLL], qk[2 * UPD_UNROLL];
#pragma unroll
                    for (int u = 0; u < UPD_UNROLL; ++u) {
                        pr[2 * u] = 1.0 - precession_pr0(xv[u][0].x, e);
                        pr[2 * u + 1] = 1.0 - precession_pr0(xv[u][0].y, e);
                    }
#pragma unroll
                    for (int q = 0; q < 2 * UPD_UNROLL; ++q) qr[q] = 1.0 - pr[q];
                    powi_uniform_n<2 * UPD_UNROLL>(pr, (unsigned)outcome, pk);
                    powi_uniform_n<2 * UPD_UNROLL>(qr, (unsigned)(e.n_meas - (double)outcome), qk);
#pragma unroll
                    for (int q = 0; q < 2 * UPD_UNROLL; ++q)
                        Lb[q] = (pr[q] >= 0.0 && pr[q] <= 1.0) ? (e.comb * pk[q]) * qk[q] : NAN;
                    batched = true;
                }
            }
#pragma unroll
            for (int u = 0; u < UPD_UNROLL; ++u) {
                const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2;
                double p0[D], p1[D];
#pragma unroll
                for (int m = 0; m < D; ++m) { p0[m] = xv[u][m].x; p1[m] = xv[u][m].y; }
                double2 wo;
                wo.x = (wi[u].x * inv_norm) * (batched ? Lb[2 * u] : model_lik<KIND, POW>(p0, e, outcome));
                wo.y = (wi[u].y * inv_norm) * (batched ? Lb[2 * u + 1] : model_lik<KIND, POW>(p1, e, outcome));
                if (nt) {
                    typedef double nt2 __attribute__((ext_vector_type(2)));
                    nt2 t;
                    t.x = wo.x;
                    t.y = wo.y;
                    __builtin_nontemporal_store(t, reinterpret_cast<nt2 *>(w_out + i));
                } else {
                    *reinterpret_cast<double2 *>(w_out + i) = wo;
                }
                acc.add(wo.x, p0);
                acc.add(wo.y, p1);
                tsum += wo.x + wo.y;
            }
        } else {
#pragma unroll
        for (int u = 0; u < UPD_UNROLL; ++u) {
            const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * VEC;
            if (VEC == 2) {
                if (i + 1 < n) {
                    typedef double nt2g __attribute__((ext_vector_type(2)));
                    double2 wi;
                    if (ONES) { wi.x = 1.0; wi.y = 1.0; }
                    else if (nt) { const nt2g t = __builtin_nontemporal_load(reinterpret_cast<const nt2g *>(w_in + i)); wi.x = t.x; wi.y = t.y; }
                    else wi = *reinterpret_cast<const double2 *>(w_in + i);
                    double p0[D], p1[D];
#pragma unroll
                    for (int m = 0; m < D; ++m) {
                        if (m < d) {
                            if (nt) {
                                const nt2g t = __builtin_nontemporal_load(reinterpret_cast<const nt2g *>(x + m * ldx + i));
                                p0[m] = t.x;
                                p1[m] = t.y;
                            } else {
                                const double2 xv = *reinterpret_cast<const double2 *>(x + m * ldx + i);
                                p0[m] = xv.x;
                                p1[m] = xv.y;
                            }
                        }
                    }
                    double2 wo;
                    wo.x = (wi.x * inv_norm) * model_lik<KIND, POW>(p0, e, outcome);
                    wo.y = (wi.y * inv_norm) * model_lik<KIND, POW>(p1, e, outcome);
                    if (nt) {
                        nt2g t;
                        t.x = wo.x;
                        t.y = wo.y;
                        __builtin_nontemporal_store(t, reinterpret_cast<nt2g *>(w_out + i));
                    } else {
                        *reinterpret_cast<double2 *>(w_out + i) = wo;
                    }
                    acc.add(wo.x, p0);
                    acc.add(wo.y, p1);
                    tsum += wo.x + wo.y;
                } else if (i < n) {
                    double p0[D];
#pragma unroll
                    for (int m = 0; m < D; ++m)
                        if (m < d) p0[m] = x[m * ldx + i];
                    const double wo = ((ONES ? 1.0 : w_in[i]) * inv_norm) * model_lik<KIND, POW>(p0, e, outcome);
                    w_out[i] = wo;
                    acc.add(wo, p0);
                    tsum += wo;
                }
            } else {
                if (i < n) {
                    double p0[D];
#pragma unroll
                    for (int m = 0; m < D; ++m)
                        if (m < d) p0[m] = x[m * ldx + i];
                    const double wo = ((ONES ? 1.0 : w_in[i]) * inv_norm) * model_lik<KIND, POW>(p0, e, outcome);
                    w_out[i] = wo;
                    acc.add(wo, p0);
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
    if (ro.tile_sums) {                          // uniform
        // the workgroup that took the last tile zeroes the entries up to the end of that tile's 4096-particle chunk, so that
        // the chunk-sum scan reads whole chunks without a bounds test per load (the buffer is sized in whole chunks).
        // (Outside the loop: inside it, one more live value took the d = 16 kernel from 168 to 169 VGPRs -- two waves
        //  per SIMD instead of three, 35 -> 55 us.)
        static_assert(4096 % TILE == 0, "tiles per chunk");
        constexpr int64_t PER_CHUNK = 4096 / TILE * QSMC_WAVES_PER_BLOCK;
        const int64_t last = (n - 1) / TILE;
        if ((int64_t)blockIdx.x == last % (int64_t)gridDim.x) {
            const int64_t first = (last + 1) * QSMC_WAVES_PER_BLOCK;
            const int64_t end = (first + PER_CHUNK - 1) / PER_CHUNK * PER_CHUNK;
            for (int64_t k = first + threadIdx.x; k < end; k += QSMC_BLOCK) ro.tile_sums[k] = 0.0;
        }
    }
    block_publish<UpdAcc<DMOM>::NS>(acc.s, acc.mn, ro);
}

// ---------------------------------------------------------------------------------------------
// Round 4: the tomography update reads the rows it needs.  TomographyModel.likelihood (tomography/models.py:211-226) is
// pr1 = clip(sum_i meas_i x_i, 0, 1): a measurement vector with NNZ nonzero entries touches NNZ of the d = 16 rows --
// and the measurements of a tomography experiment are sparse by construction: a Pauli measurement (I + P) / 2 is
// e_0 + e_P in the Pauli basis (RandomPauliHeuristic, tomography/expdesign.py:134-160; SURVEY config 5): NNZ = 2.
// k_update_fused<TOMOGRAPHY> loads all 16 rows (144 B per particle: 35 us at N = 1.25e6, 0.63 of the roofline) to
// multiply 14 of them by zero; this kernel loads w and the NNZ rows (16 + 8 NNZ = 32 B per particle).  The skipped terms
// are +-0 and x + (+-0) = x in IEEE arithmetic, so for finite particles the sum -- added in ascending i as before --
// has the same bits (a NaN in a coordinate the measurement does not look at no longer poisons the weight; the
// reference's 0 * NaN would).  The weighted moments of d > 4 are a pass of their own (k_moments_mfma), so nothing
// else in the update wants the other rows.  Same tiles, tile sums and partials as k_update_fused: everything behind
// it (reduction, chunk prefix, speculative counts, resample) is unchanged.  Full tiles issue all their loads first.
// ---------------------------------------------------------------------------------------------
template <int NNZ, bool ONES>
__global__ __launch_bounds__(QSMC_BLOCK) void k_update_tomo(
    const double *__restrict__ x, int64_t ldx, int64_t n, const double *__restrict__ w_in,
    double *__restrict__ w_out, double prev_norm, ExpArgs e, int64_t outcome, ReduceOut ro) {
    constexpr int64_t TILE = (int64_t)QSMC_BLOCK * 2 * UPD_UNROLL;
    UpdAcc<0> acc;
    acc.init();
    const double inv_norm = 1.0 / prev_norm;
    const double *row[NNZ];
    double mv[NNZ];
#pragma unroll
    for (int j = 0; j < NNZ; ++j) {
        row[j] = x + (int64_t)e.nz_idx[j] * ldx;
        mv[j] = e.meas[e.nz_idx[j]];
    }
    auto lik = [&](const double *xs) -> double {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NNZ; ++j) s += mv[j] * xs[j];
        const double pr1 = fmin(fmax(s, 0.0), 1.0);
        return two_outcome(1.0 - pr1, outcome);
    };
    for (int64_t base = (int64_t)blockIdx.x * TILE; base < n; base += (int64_t)gridDim.x * TILE) {
        double tsum = 0.0;
        if (base + TILE <= n) {
            double2 wi[UPD_UNROLL], xv[UPD_UNROLL][NNZ];
#pragma unroll
            for (int u = 0; u < UPD_UNROLL; ++u) {
                const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2;
                if (ONES) { wi[u].x = 1.0; wi[u].y = 1.0; } else wi[u] = *reinterpret_cast<const double2 *>(w_in + i);
#pragma unroll
                for (int j = 0; j < NNZ; ++j) xv[u][j] = *reinterpret_cast<const double2 *>(row[j] + i);
            }
#pragma unroll
            for (int u = 0; u < UPD_UNROLL; ++u) {
                const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2;
                double a0[NNZ], a1[NNZ];
#pragma unroll
                for (int j = 0; j < NNZ; ++j) { a0[j] = xv[u][j].x; a1[j] = xv[u][j].y; }
                double2 wo;
                wo.x = (wi[u].x * inv_norm) * lik(a0);
                wo.y = (wi[u].y * inv_norm) * lik(a1);
                *reinterpret_cast<double2 *>(w_out + i) = wo;
                acc.add(wo.x, nullptr);
                acc.add(wo.y, nullptr);
                tsum += wo.x + wo.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < UPD_UNROLL; ++u) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2 + h;
                    if (i < n) {
                        double a[NNZ];
#pragma unroll
                        for (int j = 0; j < NNZ; ++j) a[j] = row[j][i];
                        const double wo = ((ONES ? 1.0 : w_in[i]) * inv_norm) * lik(a);
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

// ---------------------------------------------------------------------------------------------
// K data in ONE pass (batch_update between two ESS checks, smc.py:459-487): the reference
// renormalises after every datum, but the normaliser is a scalar, so
//     w_K = w_0 * prod_k L_k / S_K,   S_k = sum_i w_0,i prod_{j<=k} L_j,i,
// normalization_record[k] = S_k / S_{k-1} and n_ess after datum k = S_k^2 / Q_k (Q_k the sum of
// squares).  The cloud crosses HBM once per K data instead of K times; per datum the kernel keeps
// [S_k, Q_k, #bad_k] so the host can replay every guard / record of the reference.
// ---------------------------------------------------------------------------------------------
constexpr int MULTI_KMAX = 8;
struct MultiArgs {
    int k;
    ExpArgs e[MULTI_KMAX];
    int64_t outcome[MULTI_KMAX];
    // SimplePrecession windows (round 5): the likelihood of datum k as fma(lb, pr0, la) -- (la, lb) = (0, 1) for outcome 0,
    // (1, -1) otherwise: pr0 and 1 - pr0 to the bit, without the two v_cndmask of a select on a double -- and what bounds
    // every datum's cosine argument for a particle: |t_k (omega - w_k) / 2| <= (|omega| + wabs_max) half_tmax
    unsigned int outcome_mask;      // bit k: outcome k != 0 (la = 1, lb = -1)
    double half_tmax, wabs_max;
};

constexpr int MULTI_PER_THREAD = 8;
template <int KIND, bool POW>
__global__ __launch_bounds__(QSMC_BLOCK) void k_update_multi(
    const double *__restrict__ x, int64_t ldx, int64_t n, const double *__restrict__ w_in,
    double *__restrict__ w_out, double prev_norm, MultiArgs ma, ReduceOut ro) {
    constexpr int D = Model<KIND>::D;
    constexpr int DMOM = D <= 4 ? D : 0;
    constexpr int NS = 3 * MULTI_KMAX + DMOM + DMOM * (DMOM + 1) / 2;
    const int d = (KIND == QSMC_MODEL_TOMOGRAPHY) ? ma.e[0].d : D;
    double s[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) s[q] = 0.0;
    double mn = INFINITY;
    const double inv_norm = 1.0 / prev_norm;
    // tiles of 2048 consecutive particles (eight per thread), as in k_update_fused with 16-byte loads: the sum of the final
    // weights per tile and wave goes to ro.tile_sums, so that a resample after the window starts from them (and from the
    // chunk-sum prefix the reducing launch forms beside the reduction) instead of reading the weights once more
    constexpr int64_t TILE = (int64_t)QSMC_BLOCK * MULTI_PER_THREAD;
    for (int64_t base = (int64_t)blockIdx.x * TILE; base < n; base += (int64_t)gridDim.x * TILE) {
        double tsum = 0.0;
        bool done = false;
        if constexpr (KIND == QSMC_MODEL_PRECESSION && !POW) {
            // SimplePrecession, a full tile (round 5): the loop nest TRANSPOSED -- the lane's eight particles are loaded first
            // (all sixteen loads in flight), then datum by datum over the eight.  What the datum-inside-particle order cost
            // (ISA of round 4's kernel, ~46 VALU instructions per (particle, datum), 72 % VALU-busy): the window's 4 K
            // uniforms (t, w, the outcome's two coefficients) do not fit the SGPR file next to everything else and came
            // back through v_readlane, four per likelihood; cos_sq's |x| <= 1e10 test with its exec-mask save / restore and
            // branch sat around every likelihood; the outcome select was two v_cndmask.  Here a datum's uniforms are read
            // once per eight likelihoods, ONE range test covers the tile (|t_k (omega - w_k) / 2| <= (|omega| + max |w|)
            // max |t| / 2 for every k), the select is fma(lb, pr0, la) -- (0, 1) for outcome 0, (1, -1) otherwise: pr0 and
            // 1 - pr0 to the bit.  Per thread the particles still enter every sum in ascending order: the same bits as the
            // other order, weights and sums (tests: the window against the per-datum loop, and the generic path below under
            // QSMC_MULTI_GENERIC=1 bit for bit).
            if (base + TILE <= n && ma.half_tmax >= 0.0) {
                double xs[MULTI_PER_THREAD], ws[MULTI_PER_THREAD];
#pragma unroll
                for (int u = 0; u < MULTI_PER_THREAD; ++u) {
                    const int64_t i = base + (int64_t)u * QSMC_BLOCK + threadIdx.x;
                    xs[u] = x[i];
                    ws[u] = w_in ? w_in[i] : 1.0;
                }
                bool fast = true;
#pragma unroll
                for (int u = 0; u < MULTI_PER_THREAD; ++u) fast = fast && ((fabs(xs[u]) + ma.wabs_max) * ma.half_tmax <= 0.99e10);
                if (fast) {
#pragma unroll
                    for (int u = 0; u < MULTI_PER_THREAD; ++u) ws[u] = ws[u] * inv_norm;
#pragma unroll
                    for (int k = 0; k < MULTI_KMAX; ++k) {
                        if (k < ma.k) {
                            const double tk = ma.e[k].t, wk = ma.e[k].w_;
                            const bool one = (ma.outcome_mask >> k) & 1u;                  // (uniform: scalar selects)
                            const double la = one ? 1.0 : 0.0, lb = one ? -1.0 : 1.0;
#pragma unroll
                            for (int u = 0; u < MULTI_PER_THREAD; ++u) {
                                const double dw = xs[u] - wk;
                                const double w = ws[u] * fma(lb, cos_sq_inrange(tk * dw / 2.0), la);
                                ws[u] = w;
                                s[3 * k] += w;
                                s[3 * k + 1] += w * w;
                                s[3 * k + 2] += (w >= 0.0) ? 0.0 : 1.0;
                                mn = fmin(mn, w);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < MULTI_PER_THREAD; ++u) {
                        const int64_t i = base + (int64_t)u * QSMC_BLOCK + threadIdx.x;
                        const double w = ws[u];
                        w_out[i] = w;
                        tsum += w;
                        const double wx = w * xs[u];
                        s[3 * MULTI_KMAX] += wx;
                        s[3 * MULTI_KMAX + 1] += wx * xs[u];
                    }
                    done = true;
                }
            }
        }
        if (!done) {
#pragma unroll 1
        for (int u = 0; u < MULTI_PER_THREAD; ++u) {
            const int64_t i = base + (int64_t)u * QSMC_BLOCK + threadIdx.x;
            if (i < n) {
                double p[D];
#pragma unroll
                for (int m = 0; m < D; ++m)
                    if (m < d) p[m] = x[m * ldx + i];
                double w = (w_in ? w_in[i] : 1.0) * inv_norm;
#pragma unroll
                for (int k = 0; k < MULTI_KMAX; ++k) {
                    if (k < ma.k) {
                        w = w * model_lik<KIND, POW>(p, ma.e[k], ma.outcome[k]);
                        s[3 * k] += w;
                        s[3 * k + 1] += w * w;
                        s[3 * k + 2] += (w >= 0.0) ? 0.0 : 1.0;
                        mn = fmin(mn, w);
                    }
                }
                w_out[i] = w;
                tsum += w;
                int q = 3 * MULTI_KMAX + DMOM;
#pragma unroll
                for (int m = 0; m < DMOM; ++m) {
                    const double wx = w * p[m];
                    s[3 * MULTI_KMAX + m] += wx;
#pragma unroll
                    for (int m2 = m; m2 < DMOM; ++m2) s[q++] += wx * p[m2];
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
    block_publish<NS>(s, mn, ro);
}

// ---------------------------------------------------------------------------------------------
// Round 5: the window kernel for TOMOGRAPHY with sparse measurement vectors (batch_update on config 5; smc.py:459-487 with
// tomography/models.py:211-226 inside).  k_update_multi<TOMOGRAPHY> loads all 16 rows of a particle (144 B per particle
// and window) to multiply most of them by zero K times over; a Pauli measurement touches two rows (k_update_tomo above).
// Here datum k of the window reads ITS rows: at most NZ per datum, K NZ loads per particle, of which row 0 (the trace
// coordinate, in every (I + P) / 2) and any Pauli measured twice repeat -- repeats are cache hits, so HBM sees the UNION
// of the window's rows once: 16 + 8 U bytes per particle and window (U ~ 1 + K for K distinct Paulis; 64 B at K = 5
// against 5 x 32 = 160 B for five single-datum updates and 144 B for the dense window).  A datum with fewer than NZ
// entries is padded with (row 0, coefficient +0): s + (+0 x) = s for finite x, the same bits -- as is skipping the
// rows whose coefficient is zero in the first place (k_update_tomo's argument).  The products are chained exactly as in
// k_update_multi: w_k = w_{k-1} L_k, per-datum sums [S_k, Q_k, #bad_k], the minimum over the window; same tiles, tile
// sums and partials (NS = 3 MULTI_KMAX: the d > 4 window carries no moments), so reduction, chunk prefix and resample
// behind it are unchanged -- and the weights are bitwise those of the dense window (tests: QSMC_TOMO_DENSE_UPDATE).
// K and NZ are template parameters: no uniform branch inside a particle, every load of a lane's two particles issued
// before the first product.
// ---------------------------------------------------------------------------------------------
constexpr int MULTI_TOMO_NZ = 4;
struct MultiTomoArgs {
    int32_t idx[MULTI_KMAX][MULTI_TOMO_NZ];        // rows of datum k, ascending; padding: row 0
    double mv[MULTI_KMAX][MULTI_TOMO_NZ];          // their coefficients; padding: +0
    int64_t outcome[MULTI_KMAX];
};

template <int K, int NZ>
__global__ __launch_bounds__(QSMC_BLOCK) void k_update_multi_tomo(
    const double *__restrict__ x, int64_t ldx, int64_t n, const double *__restrict__ w_in,
    double *__restrict__ w_out, double prev_norm, MultiTomoArgs ma, ReduceOut ro) {
    constexpr int NS = 3 * MULTI_KMAX;
    constexpr int64_t TILE = (int64_t)QSMC_BLOCK * MULTI_PER_THREAD;       // 2048: k_update_multi's tiles
    double s[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) s[q] = 0.0;
    double mn = INFINITY;
    const double inv_norm = 1.0 / prev_norm;
    const double *row[K][NZ];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < NZ; ++j) row[k][j] = x + (int64_t)ma.idx[k][j] * ldx;
    auto step = [&](double w, const double (&xs)[K][NZ], double &tsum) -> double {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < NZ; ++j) acc += ma.mv[k][j] * xs[k][j];
            const double pr1 = fmin(fmax(acc, 0.0), 1.0);
            w = w * two_outcome(1.0 - pr1, ma.outcome[k]);
            s[3 * k] += w;
            s[3 * k + 1] += w * w;
            s[3 * k + 2] += (w >= 0.0) ? 0.0 : 1.0;
            mn = fmin(mn, w);
        }
        tsum += w;
        return w;
    };
    for (int64_t base = (int64_t)blockIdx.x * TILE; base < n; base += (int64_t)gridDim.x * TILE) {
        double tsum = 0.0;
        if (base + TILE <= n) {
#pragma unroll 1
            for (int u = 0; u < MULTI_PER_THREAD / 2; ++u) {
                const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2;
                double2 wi, xv[K][NZ];
                if (w_in) wi = *reinterpret_cast<const double2 *>(w_in + i); else { wi.x = 1.0; wi.y = 1.0; }
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int j = 0; j < NZ; ++j) xv[k][j] = *reinterpret_cast<const double2 *>(row[k][j] + i);
                double a0[K][NZ], a1[K][NZ];
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int j = 0; j < NZ; ++j) { a0[k][j] = xv[k][j].x; a1[k][j] = xv[k][j].y; }
                double2 wo;
                wo.x = step(wi.x * inv_norm, a0, tsum);
                wo.y = step(wi.y * inv_norm, a1, tsum);
                *reinterpret_cast<double2 *>(w_out + i) = wo;
            }
        } else {
#pragma unroll 1
            for (int u = 0; u < MULTI_PER_THREAD; ++u) {
                const int64_t i = base + ((int64_t)(u >> 1) * QSMC_BLOCK + threadIdx.x) * 2 + (u & 1);
                if (i < n) {
                    double a[K][NZ];
#pragma unroll
                    for (int k = 0; k < K; ++k)
#pragma unroll
                        for (int j = 0; j < NZ; ++j) a[k][j] = row[k][j][i];
                    w_out[i] = step((w_in ? w_in[i] : 1.0) * inv_norm, a, tsum);
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
    // (sums of data past K stay zero: the host reads the first K triples)
    block_publish<NS>(s, mn, ro);
}

// ---------------------------------------------------------------------------------------------
// Experiment-design sums (bayes_risk / expected_information_gain, smc.py:553-663): for ONE
// hypothetical experiment and up to NO outcomes, in one pass over the cloud and without
// materialising L[n_o, N]:
//   S0_o = sum w L_o          (= hypothetical normalisation N[o])
//   SL_o = sum w L_o log L_o  (0 log 0 := 0)      -> N KLD = SL - S0 log S0
//   S1_o,m = sum w L_o (x_m - c_m),  S2_o,m = sum w L_o (x_m - c_m)^2   -> N var = sum_m Q_m (S2 - S1^2/S0)
// with w = w_raw / norm and c a shift (the current mean) that removes the cancellation of the
// one-pass variance.  Layout of the NS sums: [o][2 + 2 D].
// ---------------------------------------------------------------------------------------------
template <int NO>
struct HypArgs {
    int n_o;
    ExpArgs base;
    double comb[NO], log_comb[NO];
    int64_t outcome[NO];
    double shift[QSMC_MAX_D];
};

template <int KIND, int NO>
__global__ __launch_bounds__(QSMC_BLOCK) void k_hyp_sums(const double *__restrict__ x, int64_t ldx, int64_t n,
                                                         const double *__restrict__ w, double norm,
                                                         HypArgs<NO> ha, ReduceOut ro) {
    constexpr int D = Model<KIND>::D <= 4 ? Model<KIND>::D : 0;    // d > 4: normalisations and SL only
    constexpr int DD = Model<KIND>::D;
    constexpr int PER = 2 + 2 * D;
    constexpr int NS = NO * PER;
    const int d = (KIND == QSMC_MODEL_TOMOGRAPHY) ? ha.base.d : DD;
    double s[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) s[q] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double p[DD];
#pragma unroll
        for (int m = 0; m < DD; ++m)
            if (m < d) p[m] = x[m * ldx + i];
        const double wi = (w ? w[i] : 1.0) / norm;
        double c1[D > 0 ? D : 1];
#pragma unroll
        for (int m = 0; m < D; ++m) c1[m] = p[m] - ha.shift[m];
        HypPre<KIND> pre;                              // what all outcomes of this experiment share (binomial: pr1, its logs)
        pre.prepare(p, ha.base);
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            if (o < ha.n_o) {
                ExpArgs e = ha.base;
                e.comb = ha.comb[o];
                e.log_comb = ha.log_comb[o];
                double L, logL;
                pre.eval(p, e, ha.outcome[o], L, logL);
                const double wl = wi * L;
                s[o * PER] += wl;
                s[o * PER + 1] += (L > 0.0) ? wl * logL : 0.0;
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    s[o * PER + 2 + m] += wl * c1[m];
                    s[o * PER + 2 + D + m] += wl * c1[m] * c1[m];
                }
            }
        }
    }
    block_publish<NS>(s, 0.0, ro);
}

// ---------------------------------------------------------------------------------------------
// Round 4: binomial experiments, consecutive outcomes k_first, k_first + 1, ... -- the sums above with the pmfs of a pass
// WALKED instead of one exponential per (particle, outcome).  The lane-per-outcome kernel of round 3 (k_hyp_sums_lanes, removed in round 6) was VALU-bound on fast_exp (SQ counters:
// profiles/r4_*_paths_sq_counters.json): 32 outcome slots x ~46 fp64 instructions per particle.  The pmfs of consecutive
// outcomes obey  pmf(k + 1) = pmf(k) [(n - k) / (k + 1)] [p / (1 - p)]  -- along the OUTCOME axis, which the lane-per-
// outcome layout spreads over lanes; here a lane owns a particle (as in k_hyp_sums) and walks the outcomes of a pass.
// The first form (k_hyp_sums_chain, mid-round: one walk upwards from k_first with ln C and the ratios as per-outcome
// uniforms, all four sums, 13 outcomes a pass; 589 -> 2 x 176 us per 26-outcome experiment at N = 1e7) showed in its ISA
// ~20 VALU instructions per (particle, outcome) in the walk -- every `s += wl * c` is a multiply and an add (the library
// is built without contraction: the update path must match NumPy operation for operation; these sums are held to rtol
// 1e-10, not to bits), the pr1 == 1 select was four v_cndmask per outcome, the per-outcome uniforms and the uniform
// `j < n_o` guards spilled SGPRs to VGPR lanes -- and ~200 per particle and pass for two logarithms and an exponential
// that bayes_risk never uses.  The matrix cores do not come into it: on gfx950 the fp64 MFMA rate equals the fp64 vector
// rate and a 16 x 16 x 4 tile would carry 5 useful columns of 16.  (Also built and slower: tools/experiments/
// r4_hyp_sums_recurrence_via_lds.patch, the walk in the particle's lane with the pmfs handed to outcome lanes through LDS.)
// The form below:
//   * every sum is linear in the pmf, and pmf(k) = C(n, k) p^k q^(n - k): the kernel walks the GEOMETRIC sequence
//     v_j = pmf(k_ref) (p / q)^j -- one multiplication per step, no per-outcome operand -- and the host multiplies
//     the finished sums of slot j by C(n, k_ref + j) / C(n, k_ref) (chain2_collect);  ln pmf = ln C + t_j with
//     t_j = n ln q + k_j (ln p - ln q) a running sum, so sum w pmf ln pmf = scale * sum v t + ln C * sum w pmf: the
//     ln C term is added on the host as well;
//   * a pass takes its outcomes from both ends: slots 0 .. n_up - 1 upwards from k_first with p / q, the others
//     downwards from k_last with q / p.  pr1 == 0 (all mass at k = 0) and pr1 == 1 (all at k = n) are then ordinary
//     starts -- the odds are set to 0 where they would be infinite -- and nothing is selected inside the walk; an
//     invalid particle (pr1 outside [0, 1] or NaN) starts from NaN and stays there (weight 0: from 0), like SciPy's pmf;
//   * WHAT says which sums the caller uses: bayes_risk the moments (no logarithm at all; the start values are integer
//     powers for n_meas <= 64, as in binom_pmf), expected_information_gain sum w L ln L (no moments), the C ABI's
//     qsmc_hypothetical_sums both;
//   * explicit fma for the sums; slots past the pass's outcomes cost their instructions but no branch (their sums are
//     dropped on the host).
// Per (particle, outcome): 1 multiplication + 1 addition + 2 D multiply-adds (moments) / + 1 addition + 1 multiply-add
// (logarithm).  A half pass is <= 13 steps, so what an underflowing start value loses is < 1e-140 of sums that are O(1).
// ---------------------------------------------------------------------------------------------
constexpr int HYP_WHAT_LOG = 1, HYP_WHAT_MOM = 2;
struct Chain2Args {
    ExpArgs base;
    int n_up, n_dn;                // slots walked upwards from k_first / downwards from k_last (>= 1 / >= 0)
    int use_powi;                  // start values by integer powers (n_meas <= 64), else exp(ln pmf)
    unsigned k_first, k_last;
    double comb_first, comb_last;  // C(n, k_first), C(n, k_last)       (use_powi)
    double lc_first, lc_last;      // their logarithms                  (!use_powi)
    double shift[QSMC_MAX_D];
};

// ---------------------------------------------------------------------------------------------
// The two directions sit on two WAVES.  The first form of this kernel kept both walks' sums in one lane: 78 doubles for
// the 26-outcome moments pass = 256 VGPRs, two waves per SIMD, VALUs 49 % busy between dependent instructions and the
// prefetch (profiles/r4_c_paths_sq_counters.json), 120-133 us; two passes of 13 outcomes at three waves: 216 us; two or
// three particles per lane and trip: 141 / 144 us; a particle on a lane PAIR, one walk each: four waves, 65 % busy, but
// everything before the walk -- more than half of a particle's ~260 instructions -- computed twice: 174 us.  Here waves
// 2 m and 2 m + 1 of a workgroup are partners: each prepares ITS OWN 64 particles (cos^2, powers, odds: once per
// particle), keeps what its direction needs of them -- wave 2 m walks upwards, wave 2 m + 1 downwards -- and hands the
// other direction's start value, odds (and ln-pmf offset) and the shifted coordinates to the partner through LDS; after
// one barrier each wave walks ITS direction for both its own and the partner's particles.  Same instructions per
// particle (+ 2 x PAY LDS accesses per lane), half the running sums per lane: 39 for that pass = four waves per SIMD:
// 117 us, VALUs 59-69 % busy (the clock sags to ~2.0 GHz under this load).  Double-buffered exchange: one barrier per
// trip.  Columns of the partial sums: [up slots][down slots], NH x PER each.
// ---------------------------------------------------------------------------------------------
template <int KIND, int WHAT, int NH>
constexpr int chain2_lane_sums() {
    return NH * (1 + ((WHAT & HYP_WHAT_LOG) ? 1 : 0) + (((WHAT & HYP_WHAT_MOM) && Model<KIND>::D <= 4) ? 2 * Model<KIND>::D : 0));
}

template <int KIND, int WHAT, int NH>
__attribute__((amdgpu_waves_per_eu(chain2_lane_sums<KIND, WHAT, NH>() <= 39 ? 4 : (chain2_lane_sums<KIND, WHAT, NH>() <= 58 ? 3 : 2), 4)))
__global__ __launch_bounds__(QSMC_BLOCK) void k_hyp_sums_chain2(const double *__restrict__ x, int64_t ldx, int64_t n,
                                                                const double *__restrict__ w, double norm,
                                                                Chain2Args ca, ReduceOut ro) {
    constexpr bool LOG = (WHAT & HYP_WHAT_LOG) != 0, MOM = (WHAT & HYP_WHAT_MOM) != 0;
    constexpr int D = (MOM && Model<KIND>::D <= 4) ? Model<KIND>::D : 0;
    constexpr int DD = Model<KIND>::D;
    constexpr int PER = 1 + (LOG ? 1 : 0) + 2 * D;           // [S0, (St), S1[D], S2[D]]
    constexpr int NSH = NH * PER;                            // this lane's running sums: ONE direction
    constexpr int PAY = 2 + (LOG ? 2 : 0) + 2 * D;           // handed to the partner: start value, odds, (t, dl), c1[D], c2[D]
    constexpr int B1 = 1 + (LOG ? 1 : 0);
    static_assert(QSMC_WAVES_PER_BLOCK == 4, "two wave pairs per workgroup");
    __shared__ double xch[2][PAY][QSMC_BLOCK];
    __shared__ double red[QSMC_WAVES_PER_BLOCK][NSH];
    double s[NSH];
#pragma unroll
    for (int q = 0; q < NSH; ++q) s[q] = 0.0;
    const double n_meas = ca.base.n_meas;
    const double inv_norm = 1.0 / norm;
    const double kf = (double)ca.k_first, kl = (double)ca.k_last;
    const int wave = threadIdx.x / QSMC_WAVE;
    const bool down = (wave & 1) != 0;                       // (wave-uniform)
    const int ptid = threadIdx.x ^ QSMC_WAVE;                // the partner wave's lane with my lane number
    const int64_t stride = (int64_t)gridDim.x * QSMC_BLOCK;
    double pn[DD], wn = 0.0;
    {
        const int64_t i0 = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x;
#pragma unroll
        for (int m = 0; m < DD; ++m) pn[m] = 0.5;
        if (i0 < n) {
#pragma unroll
            for (int m = 0; m < DD; ++m) pn[m] = x[m * ldx + i0];
            wn = w ? w[i0] : 1.0;
        }
    }
    int buf = 0;
    // (the trip count is the workgroup's: every wave reaches every barrier; lanes past the end carry weight 0)
    for (int64_t base = (int64_t)blockIdx.x * QSMC_BLOCK; base < n; base += stride, buf ^= 1) {
        double p[DD];
#pragma unroll
        for (int m = 0; m < DD; ++m) p[m] = pn[m];
        const double wi = wn * inv_norm;
        {
            const int64_t inx = base + stride + threadIdx.x;
            if (inx < n) {
#pragma unroll
                for (int m = 0; m < DD; ++m) pn[m] = x[m * ldx + inx];
                wn = w ? w[inx] : 1.0;
            } else {
                wn = 0.0;
            }
        }
        double c1[D > 0 ? D : 1], c2[D > 0 ? D : 1];
#pragma unroll
        for (int m = 0; m < D; ++m) {
            c1[m] = p[m] - ca.shift[m];
            c2[m] = c1[m] * c1[m];
        }
        const double pr1 = hyp_pr1<KIND>(p, ca.base);
        const double qr1 = 1.0 - pr1;
        const bool valid = pr1 >= 0.0 && pr1 <= 1.0;
        const double pq = pr1 * qr1;                          // (odds p / q and q / p from ONE reciprocal, 1 / (p q): seed + two Newton steps, ~2 ulp -- a walk of 12 steps stays
        // inside 1e-14; two IEEE divisions were 28 of a particle's ~260 instructions.  0 where a walk starts on all of the mass
        // or on none and must stay at 0: p or q exactly 0, and p below 1e-290, where every pmf but pmf(0) is below 1e-290)
        const bool odds = valid && pq > 1.0e-290;
        double rc = __builtin_amdgcn_rcp(odds ? pq : 1.0);
        rc = fma(fma(-pq, rc, 1.0), rc, rc);
        rc = fma(fma(-pq, rc, 1.0), rc, rc);
        const double step = odds ? pr1 * (pr1 * rc) : 0.0;
        const double istep = odds ? qr1 * (qr1 * rc) : 0.0;
        double lp = 0.0, lq = 0.0;
        if (LOG || !ca.use_powi) {                            // (the second condition is uniform)
            lp = (valid && pr1 > 0.0) ? fast_log(pr1) : 0.0;
            lq = (valid && pr1 < 1.0) ? fast_log1m(pr1) : 0.0;
        }
        const double dl = lp - lq;
        const double tu = n_meas * lq + kf * dl, td = n_meas * lq + kl * dl;
        double su0, sd0;
        if (ca.use_powi) {
            su0 = (ca.comb_first * powi_uniform(pr1, ca.k_first)) * powi_uniform(qr1, (unsigned)n_meas - ca.k_first);
            sd0 = (ca.comb_last * powi_uniform(pr1, ca.k_last)) * powi_uniform(qr1, (unsigned)n_meas - ca.k_last);
        } else {
            const bool inside = pr1 > 0.0 && pr1 < 1.0;
            const double edge_u = pr1 == 0.0 ? (kf == 0.0 ? 1.0 : 0.0) : (kf == n_meas ? 1.0 : 0.0);
            const double edge_d = pr1 == 0.0 ? (kl == 0.0 ? 1.0 : 0.0) : (kl == n_meas ? 1.0 : 0.0);
            su0 = inside ? fast_exp(ca.lc_first + tu) : edge_u;
            sd0 = inside ? fast_exp(ca.lc_last + td) : edge_d;
        }
        const double bad = wi == 0.0 ? 0.0 : NAN;
        const double cu = valid ? wi * su0 : bad, cd = valid ? wi * sd0 : bad;
        // mine: this wave's direction of my particle; theirs: the other direction, for the partner
        double cur[2], stp[2], t[2], dls[2], a1[2][D > 0 ? D : 1], a2[2][D > 0 ? D : 1];
        cur[0] = down ? cd : cu;
        stp[0] = down ? istep : step;
        t[0] = down ? td : tu;
        dls[0] = down ? -dl : dl;
#pragma unroll
        for (int m = 0; m < D; ++m) { a1[0][m] = c1[m]; a2[0][m] = c2[m]; }
        xch[buf][0][threadIdx.x] = down ? cu : cd;
        xch[buf][1][threadIdx.x] = down ? step : istep;
        if (LOG) {
            xch[buf][2][threadIdx.x] = down ? tu : td;
            xch[buf][3][threadIdx.x] = down ? dl : -dl;
        }
#pragma unroll
        for (int m = 0; m < D; ++m) {
            xch[buf][2 + (LOG ? 2 : 0) + m][threadIdx.x] = c1[m];
            xch[buf][2 + (LOG ? 2 : 0) + D + m][threadIdx.x] = c2[m];
        }
        __syncthreads();
        cur[1] = xch[buf][0][ptid];
        stp[1] = xch[buf][1][ptid];
        t[1] = LOG ? xch[buf][LOG ? 2 : 0][ptid] : 0.0;
        dls[1] = LOG ? xch[buf][LOG ? 3 : 0][ptid] : 0.0;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            a1[1][m] = xch[buf][2 + (LOG ? 2 : 0) + m][ptid];
            a2[1][m] = xch[buf][2 + (LOG ? 2 : 0) + D + m][ptid];
        }
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            double *a = s + j * PER;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a[0] += cur[u];
                if (LOG) {
                    a[1] = fma(cur[u], t[u], a[1]);
                    t[u] += dls[u];
                }
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    a[B1 + m] = fma(cur[u], a1[u][m], a[B1 + m]);
                    a[B1 + D + m] = fma(cur[u], a2[u][m], a[B1 + D + m]);
                }
                cur[u] *= stp[u];
            }
        }
    }
    // partial sums of the workgroup: waves 0, 2 hold the upward slots, 1, 3 the downward ones
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    // (one sum at a time, reduced and parked: with the 39 shuffle trees of the 26-outcome moments pass interleaved by the
    //  scheduler this epilogue spilled 16 doubles -- 144 B of scratch per lane in a kernel held to 128 VGPRs for four
    //  waves per SIMD; never in the walk, but scratch all the same)
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
        const double t = wave_sum(s[k]);
        if (lane == 0) red[wave][k] = t;
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= 2 * NSH; k += QSMC_BLOCK) {
        double tsum = 0.0;
        if (k < 2 * NSH) {
            const int dir = k / NSH, kk = k - dir * NSH;
            tsum = red[dir][kk] + red[dir + 2][kk];
        }
        ro.partials[(size_t)k * gridDim.x + blockIdx.x] = tsum;       // (entry 2 NSH: the unused minimum slot)
    }
}

// mode 0: w_out = (w_in / norm) * L   (generic-model slow path)
// mode 1: w_out = clip(w_in / norm, 0, 1)   (negative-weight guard)
// mode 2: w_out = w_in / norm               (materialise; stats still produced)
// mode 3: stats of w_in / norm only (no store)
template <int MODE>
__global__ __launch_bounds__(QSMC_BLOCK) void k_weights_pass(const double *__restrict__ L, int64_t n,
                                                             const double *__restrict__ w_in,
                                                             double *__restrict__ w_out, double norm,
                                                             ReduceOut ro) {
    UpdAcc<0> acc;
    acc.init();
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double w = w_in[i] / norm;
        if (MODE == 0) w = w * L[i];
        if (MODE == 1 && w == w) w = fmin(fmax(w, 0.0), 1.0);   // np.clip keeps NaN as NaN
        if (MODE != 3 && MODE != 4) w_out[i] = w;
        if (MODE == 4) {
            // est_entropy (distributions.py:457-464): -sum_{w > 0} w log w, carried in the sumsq slot
            acc.s[0] += w;
            acc.s[1] += w > 0.0 ? -(w * fast_log(w)) : 0.0;
            acc.mn = fmin(acc.mn, w);
        } else {
            acc.add(w, nullptr);
        }
    }
    block_publish<3>(acc.s, acc.mn, ro);
}

