// qsmc_kernels.hip -- gfx950 kernels + C ABI (include/qsmc.h) for the SMC hot path.
//
// All kernels are HBM-bound streaming passes (elementwise + reductions + one scan + one gather);
// none is matmul-shaped at d <= 4, so there is no MFMA here.  Design points:
//   * SoA particle layout -> every global access is a unit-stride wave-wide load/store;
//   * two doubles (16 B) per lane per access where alignment allows (template VEC = 2);
//   * reductions are two-level and deterministic: per-thread registers -> wave64 shuffle tree ->
//     LDS across the 4 waves -> one partial per workgroup -> a one-workgroup finalize kernel that
//     sums the partials in index order (bitwise reproducible for a given n);
//   * weights stay unnormalised in HBM; the normaliser is a scalar folded into the next read.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <rocprim/rocprim.hpp>      // device radix sort for the posterior read-outs (a plain library op)

#include "qsmc_device.h"

using namespace qsmc;

// =============================================================================================
// context
// =============================================================================================
struct qsmc_ctx {
    int device;
    double *partials;      // device scratch for per-workgroup partial sums
    size_t partials_cap;   // in doubles
    double *scratch;       // device scratch (scan partials, small outputs)
    size_t scratch_cap;
    double *pinned;        // host pinned staging for small read-backs
    size_t pinned_cap;
    long long *counter;    // device int64 counter (failed-particle count)
    double *red_out;       // device [REDUCE_OUT_MAX] totals of the last grid reduction
    double *mapped;        // pinned host memory the reducing workgroup writes directly ...
    double *mapped_dev;    // ... and its device alias
    unsigned long long *flag;      // pinned host word: sequence number of the last completed reduction
    unsigned long long *flag_dev;  // its device alias
    unsigned long long seq;        // last sequence number handed to a reducing launch
    double *rs_offsets;            // resampler: chunk offsets (own buffer: survives other calls' scratch use)
    size_t rs_offsets_cap;
    size_t count_lds_granted;      // dynamic LDS already opted into for k_bucket_count on this device
    size_t topup_lds_granted;      // ... and for k_bucket_counts
    unsigned long long *gbar;      // device: [0] arrival counter of the count kernel's barriers (only ever grows), [1] its
                                   // timeouts; [2], [3] arrivals / departures of the redraw kernel's self-resetting barrier
    unsigned long long gbar_base;  // host shadow: arrivals handed out so far
    void *sort_tmp;                // rocPRIM temporary storage + key/value staging for qsmc_argsort
    size_t sort_tmp_cap;           // in bytes
    double *tile_sums;             // sum of w' per update-kernel tile, written by the last qsmc_update_fused
    size_t tile_sums_cap;
    struct {
        unsigned long long gen;    // counts qsmc_update_fused calls; the caller mirrors it as a token
        const double *w;           // the w_out those sums describe
        int64_t n;
        int tile;                  // particles per tile
        unsigned long long armed;  // token handed in by qsmc_lw_use_update_sums for the next resample (0 = none)
    } ts;
    struct {                       // weight-only prefix of a resample already queued (qsmc_lw_resample_prepare)
        int valid;
        const double *w;
        int64_t n_in, n_out;
        double norm;
        uint64_t seed, epoch;
        hipStream_t stream;
    } prep;
    unsigned int *iscratch; // device integer scratch for the bucketed resampler
    size_t iscratch_cap;    // in bytes
    double *cdf_scratch;    // device CDF, materialised only for the direct sampler / global redraws
    size_t cdf_cap;         // in doubles
    int profiling;
    hipEvent_t *prof_ev;   // QSMC_PROF_CAP (start, stop) pairs, created on first qsmc_set_profiling(1)
    int prof_n;            // profiled launches since the last qsmc_profile_read / set_profiling
    unsigned char *prof_tag;   // which kernel each ring entry timed (QSMC_PROF_*)
    int prof_stride;           // time every prof_stride-th launch of a tag (events cost ~10 us of queue drain each)
    unsigned prof_seen[4];     // launches seen per tag
    char hip_err[256];
};

#define HIP_TRY(h, expr)                                                                        \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            if (h) snprintf((h)->hip_err, sizeof((h)->hip_err), "%s: %s", #expr,                \
                            hipGetErrorString(e__));                                            \
            return QSMC_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

constexpr int REDUCE_OUT_MAX = 192;
constexpr int QSMC_PROF_CAP = 4096;

static int ensure_partials(qsmc_ctx *h, size_t n) {
    if (h->partials_cap >= n) return QSMC_OK;
    if (h->partials) HIP_TRY(h, hipFree(h->partials));
    h->partials = nullptr;
    h->partials_cap = 0;
    HIP_TRY(h, hipMalloc(&h->partials, n * sizeof(double)));
    h->partials_cap = n;
    return QSMC_OK;
}

static int ensure_scratch(qsmc_ctx *h, size_t n) {
    if (h->scratch_cap >= n) return QSMC_OK;
    if (h->scratch) HIP_TRY(h, hipFree(h->scratch));
    h->scratch = nullptr;
    h->scratch_cap = 0;
    HIP_TRY(h, hipMalloc(&h->scratch, n * sizeof(double)));
    h->scratch_cap = n;
    return QSMC_OK;
}

static int ensure_rs_offsets(qsmc_ctx *h, size_t n) {
    if (h->rs_offsets_cap >= n) return QSMC_OK;
    if (h->rs_offsets) HIP_TRY(h, hipFree(h->rs_offsets));
    h->rs_offsets = nullptr;
    h->rs_offsets_cap = 0;
    HIP_TRY(h, hipMalloc(&h->rs_offsets, n * sizeof(double)));
    h->rs_offsets_cap = n;
    return QSMC_OK;
}

static int ensure_tile_sums(qsmc_ctx *h, size_t n) {
    if (h->tile_sums_cap >= n) return QSMC_OK;
    if (h->tile_sums) HIP_TRY(h, hipFree(h->tile_sums));
    h->tile_sums = nullptr;
    h->tile_sums_cap = 0;
    HIP_TRY(h, hipMalloc(&h->tile_sums, n * sizeof(double)));
    h->tile_sums_cap = n;
    return QSMC_OK;
}

static int ensure_iscratch(qsmc_ctx *h, size_t bytes) {
    if (h->iscratch_cap >= bytes) return QSMC_OK;
    if (h->iscratch) HIP_TRY(h, hipFree(h->iscratch));
    h->iscratch = nullptr;
    h->iscratch_cap = 0;
    HIP_TRY(h, hipMalloc(&h->iscratch, bytes));
    h->iscratch_cap = bytes;
    return QSMC_OK;
}

static int ensure_cdf(qsmc_ctx *h, size_t n) {
    if (h->cdf_cap >= n) return QSMC_OK;
    if (h->cdf_scratch) HIP_TRY(h, hipFree(h->cdf_scratch));
    h->cdf_scratch = nullptr;
    h->cdf_cap = 0;
    HIP_TRY(h, hipMalloc(&h->cdf_scratch, n * sizeof(double)));
    h->cdf_cap = n;
    return QSMC_OK;
}

static int ensure_pinned(qsmc_ctx *h, size_t n) {
    if (h->pinned_cap >= n) return QSMC_OK;
    if (h->pinned) HIP_TRY(h, hipHostFree(h->pinned));
    h->pinned = nullptr;
    h->pinned_cap = 0;
    HIP_TRY(h, hipHostMalloc(&h->pinned, n * sizeof(double), hipHostMallocDefault));
    h->pinned_cap = n;
    return QSMC_OK;
}

// device -> host read-back of a few doubles; synchronises the stream
static int read_back(qsmc_ctx *h, const double *dev, double *host, size_t n, hipStream_t s) {
    int rc = ensure_pinned(h, n);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->pinned, dev, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    memcpy(host, h->pinned, n * sizeof(double));
    return QSMC_OK;
}

static inline int grid_for(int64_t n, int per_block) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > QSMC_GRID_CAP) g = QSMC_GRID_CAP;
    return (int)g;
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

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
    double *partials;        // [grid][NS + 1]
    double *out_dev;         // [NS + 1] device (always written)
    double *out_mapped;      // [NS + 1] device alias of pinned host memory (nullable)
    double *stats4;          // optional caller buffer in qsmc_update_stats_t order (nullable)
    unsigned long long *flag;  // device alias of the pinned completion word (nullable)
    unsigned long long seq;    // value to publish there once out_mapped is complete
    const unsigned long long *failed_src;   // the resampler's failed-particle counter (device) ...
    double *failed_dst;                     // ... copied to its pinned slot by every host-visible reduction
    double *tile_sums;                      // k_update_fused only: sum of w' per TILE particles (nullable)
};

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
        ro.partials[(size_t)blockIdx.x * (NS + 1) + k] = s;
    }
}

template <int NS>
__global__ __launch_bounds__(QSMC_BLOCK) void k_reduce_partials(int nblocks, ReduceOut ro) {
    __shared__ double lds[QSMC_WAVES_PER_BLOCK * NS];
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    double m2 = INFINITY;
#pragma unroll 4
    for (int g = threadIdx.x; g < nblocks; g += QSMC_BLOCK) {
        const double *p = ro.partials + (size_t)g * (NS + 1);
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] += p[k];
        m2 = fmin(m2, p[NS]);
    }
    block_sum<NS>(acc, lds);
    m2 = block_min(m2, lds);
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
        if (ro.failed_dst) *ro.failed_dst = (double)*ro.failed_src;
        if (ro.flag) {                   // the host spins on this word instead of hipStreamSynchronize
            __threadfence_system();
            *reinterpret_cast<volatile unsigned long long *>(ro.flag) = ro.seq;
        }
    }
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
    double *__restrict__ w_out, double prev_norm, ExpArgs e, int64_t outcome, ReduceOut ro) {
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
        if (VEC == 2 && D <= 2 && base + TILE <= n) {
            // full tile: every load of the tile is issued before the first likelihood is evaluated, so a wave
            // has UPD_UNROLL x (1 + d) 16-byte loads in flight instead of 1 + d (the guarded path below
            // serialises load -> compute -> store per sub-tile because of its bounds branches)
            double2 wi[UPD_UNROLL], xv[UPD_UNROLL][D];
#pragma unroll
            for (int u = 0; u < UPD_UNROLL; ++u) {
                const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2;
                if (ONES) { wi[u].x = 1.0; wi[u].y = 1.0; } else wi[u] = *reinterpret_cast<const double2 *>(w_in + i);
#pragma unroll
                for (int m = 0; m < D; ++m) xv[u][m] = *reinterpret_cast<const double2 *>(x + m * ldx + i);
            }
#pragma unroll
            for (int u = 0; u < UPD_UNROLL; ++u) {
                const int64_t i = base + ((int64_t)u * QSMC_BLOCK + threadIdx.x) * 2;
                double p0[D], p1[D];
#pragma unroll
                for (int m = 0; m < D; ++m) { p0[m] = xv[u][m].x; p1[m] = xv[u][m].y; }
                double2 wo;
                wo.x = (wi[u].x * inv_norm) * model_lik<KIND, POW>(p0, e, outcome);
                wo.y = (wi[u].y * inv_norm) * model_lik<KIND, POW>(p1, e, outcome);
                *reinterpret_cast<double2 *>(w_out + i) = wo;
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
                    double2 wi;
                    if (ONES) { wi.x = 1.0; wi.y = 1.0; } else wi = *reinterpret_cast<const double2 *>(w_in + i);
                    double p0[D], p1[D];
#pragma unroll
                    for (int m = 0; m < D; ++m) {
                        if (m < d) {
                            const double2 xv = *reinterpret_cast<const double2 *>(x + m * ldx + i);
                            p0[m] = xv.x;
                            p1[m] = xv.y;
                        }
                    }
                    double2 wo;
                    wo.x = (wi.x * inv_norm) * model_lik<KIND, POW>(p0, e, outcome);
                    wo.y = (wi.y * inv_norm) * model_lik<KIND, POW>(p1, e, outcome);
                    *reinterpret_cast<double2 *>(w_out + i) = wo;
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
    block_publish<UpdAcc<DMOM>::NS>(acc.s, acc.mn, ro);
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
};

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
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
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
        int q = 3 * MULTI_KMAX + DMOM;
#pragma unroll
        for (int m = 0; m < DMOM; ++m) {
            const double wx = w * p[m];
            s[3 * MULTI_KMAX + m] += wx;
#pragma unroll
            for (int m2 = m; m2 < DMOM; ++m2) s[q++] += wx * p[m2];
        }
    }
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
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            if (o < ha.n_o) {
                ExpArgs e = ha.base;
                e.comb = ha.comb[o];
                e.log_comb = ha.log_comb[o];
                const double L = model_lik_rt<KIND>(p, e, ha.outcome[o]);
                const double wl = wi * L;
                s[o * PER] += wl;
                s[o * PER + 1] += (L > 0.0) ? wl * log(L) : 0.0;
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
            acc.s[1] += w > 0.0 ? -(w * log(w)) : 0.0;
            acc.mn = fmin(acc.mn, w);
        } else {
            acc.add(w, nullptr);
        }
    }
    block_publish<3>(acc.s, acc.mn, ro);
}

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

// =============================================================================================
// inclusive scan of w / norm  (three launches: chunk sums, scan of chunk sums, chunk scans)
// =============================================================================================
constexpr int SCAN_PER_LANE = 8;                                   // 8 consecutive particles per lane (64 B)
constexpr int SCAN_WAVE_CHUNK = QSMC_WAVE * SCAN_PER_LANE;         // 512 per wave
constexpr int SCAN_WAVES = 8;                                      // 512 threads scan one chunk
constexpr int SCAN_THREADS = SCAN_WAVES * QSMC_WAVE;
constexpr int SCAN_CHUNK = SCAN_WAVE_CHUNK * SCAN_WAVES;                // 4096 per workgroup

__device__ __forceinline__ double wave_inclusive_scan(double v, int lane) {
#pragma unroll
    for (int off = 1; off < QSMC_WAVE; off <<= 1) {
        const double t = __shfl_up(v, off, QSMC_WAVE);
        if (lane >= off) v += t;
    }
    return v;
}

// (the scan pipeline multiplies by 1/norm instead of dividing: fp64 division is ~25 instructions and
// the in-sampler scan is VALU-bound; every stage uses the same expression, so they agree bit for bit)
__global__ __launch_bounds__(QSMC_BLOCK) void k_chunk_sums(const double *__restrict__ w, int64_t n,
                                                           double inv_norm, double *__restrict__ sums) {
    __shared__ double lds[QSMC_WAVES_PER_BLOCK];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    double v[1] = {0.0};
#pragma unroll
    for (int u = 0; u < SCAN_CHUNK / QSMC_BLOCK; ++u) {
        const int64_t i = base + (int64_t)u * QSMC_BLOCK + threadIdx.x;
        if (i < n) v[0] += (w ? w[i] : 1.0) * inv_norm;
    }
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = v[0];
}

__device__ __forceinline__ double wave_inclusive_max(double v, int lane) {
#pragma unroll
    for (int off = 1; off < QSMC_WAVE; off <<= 1) {
        const double t = __shfl_up(v, off, QSMC_WAVE);
        if (lane >= off) v = fmax(v, t);
    }
    return v;
}

// Exclusive scan of the chunk sums, in place, plus the grand total at sums[m]  (m + 1 outputs).
// Single 1024-thread workgroup; thread t owns a contiguous run of ceil(m / 1024) entries (serial,
// in registers), the 1024 run totals are scanned with wave shuffles + 16 wave totals.  Floating-point
// tree sums are not guaranteed monotone in the index, so the result goes through an exact prefix MAX
// (no rounding) in the same pass structure; k_chunk_scan relies on monotone offsets.
constexpr int SCAN_SUMS_THREADS = 1024;
constexpr int SCAN_SUMS_MAX_PER = 16;              // m <= 16384 chunks (N <= 6.7e7) in registers

// tiles != nullptr: chunk c's sum is (tiles[c tpc] + ... + tiles[c tpc + tpc - 1]) * inv_norm -- the per-tile, per-wave
// sums the last update kernel left behind (tpc = tiles per chunk x 4 waves) -- instead of sums[c] from k_chunk_sums.
struct TileSrc {
    const double *tiles;
    int tpc;
    int64_t n_tiles;
    double inv_norm;
};

__device__ __forceinline__ double chunk_sum_in(const double *__restrict__ sums, const TileSrc &ts, int64_t c) {
    if (!ts.tiles) return sums[c];
    if (ts.tpc == 8 && (c + 1) * 8 <= ts.n_tiles) {
        // the usual case (2 tiles x 4 waves): the chunk's eight parts are one 64-byte line -> two 32-byte loads,
        // summed in index order like the loop below
        const double4 a = *reinterpret_cast<const double4 *>(ts.tiles + c * 8);
        const double4 b = *reinterpret_cast<const double4 *>(ts.tiles + c * 8 + 4);
        return (((((((a.x + a.y) + a.z) + a.w) + b.x) + b.y) + b.z) + b.w) * ts.inv_norm;
    }
    double t = 0.0;
    for (int j = 0; j < ts.tpc; ++j) {
        const int64_t k = c * ts.tpc + j;
        if (k < ts.n_tiles) t += ts.tiles[k];
    }
    return t * ts.inv_norm;
}

// One workgroup of SCAN_SUMS_THREADS: exclusive, monotone prefix of the m chunk sums; sink(i, offsets[i]) for
// i = 0 .. m (offsets[m] = total).  The sums come from `sums` or, with ts.tiles, from the update kernel's tile sums.
template <class Sink>
__device__ __forceinline__ void scan_sums_block(const double *sums, int64_t m, const TileSrc &ts, Sink sink) {
    __shared__ double wtot[SCAN_SUMS_THREADS / QSMC_WAVE];
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
    const int per = (int)((m + SCAN_SUMS_THREADS - 1) / SCAN_SUMS_THREADS);
    const int64_t i0 = (int64_t)threadIdx.x * per;
    double v[SCAN_SUMS_MAX_PER];
    double run = 0.0;
#pragma unroll
    for (int q = 0; q < SCAN_SUMS_MAX_PER; ++q) {
        v[q] = (q < per && i0 + q < m) ? chunk_sum_in(sums, ts, i0 + q) : 0.0;
        run += v[q];
    }
    // exclusive offset of this thread's run
    double inc = wave_inclusive_scan(run, lane);
    if (lane == QSMC_WAVE - 1) wtot[wave] = inc;
    __syncthreads();
    double off = inc - run;
    for (int wv = 0; wv < wave; ++wv) off += wtot[wv];
    double total = 0.0;
    for (int wv = 0; wv < SCAN_SUMS_THREADS / QSMC_WAVE; ++wv) total += wtot[wv];
    __syncthreads();
    // exclusive values of my entries, then make everything monotone with an exact prefix max
    double e[SCAN_SUMS_MAX_PER];
    double acc = off, mx = 0.0;
#pragma unroll
    for (int q = 0; q < SCAN_SUMS_MAX_PER; ++q) {
        e[q] = acc;
        acc += v[q];
        mx = fmax(mx, e[q]);
        e[q] = mx;                                  // local running max (values are >= 0)
    }
    double wmx = wave_inclusive_max(mx, lane);
    if (lane == QSMC_WAVE - 1) wtot[wave] = wmx;
    __syncthreads();
    double before = __shfl_up(wmx, 1, QSMC_WAVE);   // max over earlier lanes of this wave
    if (lane == 0) before = 0.0;
    for (int wv = 0; wv < wave; ++wv) before = fmax(before, wtot[wv]);
#pragma unroll
    for (int q = 0; q < SCAN_SUMS_MAX_PER; ++q)
        if (q < per && i0 + q < m) sink(i0 + q, fmax(e[q], before));
    if (threadIdx.x == SCAN_SUMS_THREADS - 1) {
        double gmax = 0.0;
        for (int wv = 0; wv < SCAN_SUMS_THREADS / QSMC_WAVE; ++wv) gmax = fmax(gmax, wtot[wv]);
        sink(m, fmax(total, gmax));
    }
}

__global__ __launch_bounds__(SCAN_SUMS_THREADS) void k_scan_sums(double *sums, int64_t m,
                                                                 unsigned long long *__restrict__ zero2, TileSrc ts) {
    if (zero2 && threadIdx.x < 2) zero2[threadIdx.x] = 0ull;      // the resampler's failed / retry counters (was a memset launch)
    scan_sums_block(sums, m, ts, [&](int64_t i, double v) { sums[i] = v; });
}

// Fallback for m > 16384 chunk sums (N > 6.7e7): same contract, 256-wide slabs with a carry.
__global__ __launch_bounds__(QSMC_BLOCK) void k_scan_sums_big(double *__restrict__ sums, int64_t m,
                                                              unsigned long long *__restrict__ zero2, TileSrc ts) {
    __shared__ double wave_tot[QSMC_WAVES_PER_BLOCK];
    if (zero2 && threadIdx.x < 2) zero2[threadIdx.x] = 0ull;
    __shared__ double carry_s;
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
    if (threadIdx.x == 0) carry_s = 0.0;
    __syncthreads();
    for (int64_t base = 0; base < m; base += QSMC_BLOCK) {
        const int64_t i = base + threadIdx.x;
        const double v = i < m ? chunk_sum_in(sums, ts, i) : 0.0;
        const double inc = wave_inclusive_scan(v, lane);
        double excl = __shfl_up(inc, 1, QSMC_WAVE);
        if (lane == 0) excl = 0.0;
        if (lane == QSMC_WAVE - 1) wave_tot[wave] = inc;
        __syncthreads();
        double off = carry_s;
        for (int wv = 0; wv < wave; ++wv) off += wave_tot[wv];
        if (i < m) sums[i] = off + excl;
        __syncthreads();
        if (threadIdx.x == QSMC_BLOCK - 1) carry_s = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[m] = carry_s;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = 0.0;
    __syncthreads();
    for (int64_t base = 0; base <= m; base += QSMC_BLOCK) {      // exact prefix max over sums[0..m]
        const int64_t i = base + threadIdx.x;
        const double v = i <= m ? sums[i] : 0.0;
        const double mx = wave_inclusive_max(v, lane);
        if (lane == QSMC_WAVE - 1) wave_tot[wave] = mx;
        __syncthreads();
        double run = carry_s;
        for (int wv = 0; wv < wave; ++wv) run = fmax(run, wave_tot[wv]);
        const double out = fmax(mx, run);
        if (i <= m) sums[i] = out;
        __syncthreads();
        if (threadIdx.x == QSMC_BLOCK - 1) carry_s = out;
        __syncthreads();
    }
}

// Per-chunk scan.  offsets[] has chunks + 1 monotone entries (exclusive offsets + total).
// Lane l of wave v owns the 8 consecutive particles [512 v + 8 l, +8): a serial running sum in
// registers (monotone by construction), one wave scan of the lane totals, 8 wave totals through LDS.
// Every value is clamped into its wave's [lo, hi] offset window and the lanes' last values go through
// an exact prefix max, so the CDF is non-decreasing everywhere (searchsorted on it is well defined)
// while differing from the sequential np.cumsum only by rounding.  The last entry of a wave's
// 512-particle segment is DEFINED as the window top hi, and the chunk's last entry as offsets[c + 1]
// (equal in exact arithmetic), so chunk edges and CDF entries are one and the same numbers whether or
// not the CDF is ever written to HBM, and each lane knows its predecessor's value without a barrier.
// The first 512 threads scan chunk c; every thread of the workgroup must call (one barrier inside).
// store(j, value, prev) receives the chunk-local index, the entry and the entry before it (the chunk's
// lower edge for j = 0); calls are made wave-uniformly (`live` = the entry exists).
template <class Store>
__device__ __forceinline__ void chunk_scan_block(const double *__restrict__ w, int64_t n, double inv_norm,
                                                 const double *__restrict__ offsets, int64_t c,
                                                 double *wave_tot, Store store) {
    const bool act = threadIdx.x < SCAN_THREADS;
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = (threadIdx.x / QSMC_WAVE) & (SCAN_WAVES - 1);
    const int j0 = wave * SCAN_WAVE_CHUNK + lane * SCAN_PER_LANE;          // chunk-local index of v[0]
    const int64_t i0 = c * SCAN_CHUNK + j0;
    double v[SCAN_PER_LANE];
    double excl = 0.0;
    if (act) {
        if (!w) {
#pragma unroll
            for (int k = 0; k < SCAN_PER_LANE; ++k) v[k] = i0 + k < n ? inv_norm : 0.0;
        } else if (i0 + SCAN_PER_LANE <= n && ((uintptr_t)w & 31) == 0) {
#pragma unroll
            for (int k = 0; k < SCAN_PER_LANE; k += 4) {
                const double4 t = *reinterpret_cast<const double4 *>(w + i0 + k);
                v[k] = t.x * inv_norm; v[k + 1] = t.y * inv_norm; v[k + 2] = t.z * inv_norm; v[k + 3] = t.w * inv_norm;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SCAN_PER_LANE; ++k) v[k] = i0 + k < n ? w[i0 + k] * inv_norm : 0.0;
        }
#pragma unroll
        for (int k = 1; k < SCAN_PER_LANE; ++k) v[k] += v[k - 1];
        const double inc = wave_inclusive_scan(v[SCAN_PER_LANE - 1], lane);
        excl = __shfl_up(inc, 1, QSMC_WAVE);
        if (lane == 0) excl = 0.0;
        if (lane == QSMC_WAVE - 1) wave_tot[wave] = inc;
    }
    __syncthreads();
    if (!act) return;
    const double blo = c <= 0 ? 0.0 : offsets[c], bhi = offsets[c + 1];
    const int len = (int)((n - c * SCAN_CHUNK) < SCAN_CHUNK ? (n - c * SCAN_CHUNK) : SCAN_CHUNK);
    double lo = blo;
    for (int wv = 0; wv < wave; ++wv) lo = fmin(lo + wave_tot[wv], bhi);
    const double hi = (wave == SCAN_WAVES - 1) ? bhi : fmin(lo + wave_tot[wave], bhi);
#pragma unroll
    for (int k = 0; k < SCAN_PER_LANE; ++k) v[k] = fmin(fmax(lo + (excl + v[k]), lo), hi);
    const double m = wave_inclusive_max(v[SCAN_PER_LANE - 1], lane);
    double prev = __shfl_up(m, 1, QSMC_WAVE);
    if (lane == 0) prev = lo;                      // == the previous wave's (forced) last entry, or the chunk's lower edge
#pragma unroll
    for (int k = 0; k < SCAN_PER_LANE; ++k) {
        const int j = j0 + k;
        double a = fmax(v[k], prev);
        if (j == len - 1) a = bhi;
        else if (lane == QSMC_WAVE - 1 && k == SCAN_PER_LANE - 1) a = hi;
        store(j, a, prev, j < len);
        prev = a;
    }
}

struct StoreGlobal {
    double *cdf;                                   // + chunk base
    __device__ __forceinline__ void operator()(int j, double v, double, bool live) const {
        if (live) cdf[j] = v;
    }
};

// Materialise the CDF.  gate != nullptr: do nothing unless *gate > 0 (the bucketed resampler only
// needs the global CDF when some particle has to redraw a global ancestor).
__global__ __launch_bounds__(SCAN_THREADS) void k_chunk_scan(const double *__restrict__ w, int64_t n,
                                                             double inv_norm, const double *__restrict__ offsets,
                                                             double *__restrict__ cdf,
                                                             const unsigned long long *__restrict__ gate) {
    __shared__ double wave_tot[SCAN_WAVES];
    if (gate && *gate == 0ull) return;
    chunk_scan_block(w, n, inv_norm, offsets, (int64_t)blockIdx.x, wave_tot,
                     StoreGlobal{cdf + (int64_t)blockIdx.x * SCAN_CHUNK});
}

// =============================================================================================
// Liu-West pieces
// =============================================================================================
// upper bound: number of entries <= u, clamped to n - 1
__device__ __forceinline__ int64_t search_right(const double *__restrict__ cdf, int64_t n, double u) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo < n - 1 ? lo : n - 1;
}

__global__ __launch_bounds__(QSMC_BLOCK) void k_ancestors(const double *__restrict__ cdf, int64_t n_in,
                                                          const double *__restrict__ u, int64_t n_out,
                                                          int64_t *__restrict__ js) {
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n_out;
         i += (int64_t)gridDim.x * QSMC_BLOCK)
        js[i] = search_right(cdf, n_in, u[i]);
}

struct LWArgs {
    double a;
    double mean[QSMC_MAX_D];
    double S[QSMC_MAX_D * QSMC_MAX_D];   // row-major d x d (already times h)
};

__global__ __launch_bounds__(QSMC_BLOCK) void k_centres(const double *__restrict__ x_in, int64_t ldx_in,
                                                        int d, const int64_t *__restrict__ js,
                                                        int64_t n_out, double a, LWArgs lw,
                                                        double *__restrict__ mus, int64_t ld_mus) {
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n_out;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t j = js[i];
        for (int m = 0; m < d; ++m)
            mus[m * ld_mus + i] = a * x_in[m * ldx_in + j] + (1.0 - a) * lw.mean[m];   // :325
    }
}

__global__ __launch_bounds__(QSMC_BLOCK) void k_perturb(int kind, int d, double min_freq, int postselect,
                                                        const double *__restrict__ mus, int64_t ld_mus,
                                                        const int64_t *__restrict__ idxs, int64_t k,
                                                        int centre_by_idx, LWArgs lw,
                                                        const double *__restrict__ z, int64_t ldz,
                                                        double *__restrict__ x_out, int64_t ldx_out,
                                                        uint8_t *__restrict__ valid) {
    for (int64_t r = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; r < k;
         r += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t dst = idxs ? idxs[r] : r;
        const int64_t c = centre_by_idx ? dst : r;
        double p[QSMC_MAX_D];
        for (int m = 0; m < d; ++m) {
            double s = 0.0;                     // (S @ z)[m, r], summed in column order like np.dot
            for (int q = 0; q < d; ++q) s += lw.S[m * d + q] * z[q * ldz + r];
            p[m] = mus[m * ld_mus + c] + s;
            x_out[m * ldx_out + dst] = p[m];
        }
        valid[r] = (!postselect || model_valid(kind, p, min_freq)) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Output placement.  Single GPU: slot o -> column o of the SoA cloud.  Sharded: this
// rank produces the finished particles for EVERY destination rank and they leave by one
// all_to_all, so rows must be grouped by destination, AoS.  The bucketed sampler emits slots sorted
// by ancestor chunk; dealing them to destinations round-robin (skipping a destination once its
// quota is full -- exact quotas, closed form below) gives every destination an even, stratified
// share of all chunks, so the shards stay statistically exchangeable.
// ---------------------------------------------------------------------------------------------
#define QSMC_MAX_DEST 16
struct OutPlace {
    int n_dest;                            // 0: identity placement
    int64_t ld_m, ld_s;                    // element (m, row) lives at x_out[m * ld_m + row * ld_s]
    int order[QSMC_MAX_DEST];              // destination ids by ascending quota
    int64_t quota[QSMC_MAX_DEST];          // ascending quotas c_(0) <= ... <= c_(G-1)
    int64_t seg_start[QSMC_MAX_DEST + 1];  // first slot of dealing segment s (rounds c_(s-1) .. c_(s)-1)
    int64_t dest_base[QSMC_MAX_DEST];      // first row of destination r
};

__device__ __forceinline__ int64_t place_row(const OutPlace &pl, int64_t o) {
    if (pl.n_dest == 0) return o;
    int s = 0;
    while (s + 1 < pl.n_dest && o >= pl.seg_start[s + 1]) ++s;
    const int active = pl.n_dest - s;
    const int64_t rel = o - pl.seg_start[s];
    const int64_t round = (s ? pl.quota[s - 1] : 0) + rel / active;
    const int dest = pl.order[s + (int)(rel % active)];
    return pl.dest_base[dest] + round;
}

// One-launch device-RNG resample.
__global__ __launch_bounds__(QSMC_BLOCK) void k_resample_philox(
    int kind, int d, double min_freq, int postselect, const double *__restrict__ x_in, int64_t ldx_in,
    int64_t n_in, const double *__restrict__ cdf, LWArgs lw, int64_t n_out, uint32_t k0, uint32_t k1,
    uint32_t epoch, int maxiter, double *__restrict__ x_out, OutPlace pl,
    unsigned long long *__restrict__ n_failed) {
    unsigned long long failed = 0;
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n_out;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double p[QSMC_MAX_D];
        bool ok = false;
        for (int round = 0; round < maxiter && !ok; ++round) {
            PhiloxStream rng{(uint64_t)i, (epoch << 16) | (uint32_t)round, k0, k1};
            double u, unused;
            rng.uniforms(0, u, unused);
            const int64_t j = search_right(cdf, n_in, u);
            double zz[QSMC_MAX_D];
            for (int q = 0; q < d; q += 2) {
                double z0, z1;
                rng.normals(1 + (q >> 1), z0, z1);
                zz[q] = z0;
                if (q + 1 < d) zz[q + 1] = z1;
            }
            for (int m = 0; m < d; ++m) {
                double s = 0.0;
                for (int q = 0; q < d; ++q) s += lw.S[m * d + q] * zz[q];
                p[m] = (lw.a * x_in[m * ldx_in + j] + (1.0 - lw.a) * lw.mean[m]) + s;
            }
            ok = !postselect || model_valid(kind, p, min_freq);
        }
        const int64_t row = place_row(pl, i);
        for (int m = 0; m < d; ++m) x_out[m * pl.ld_m + row * pl.ld_s] = p[m];
        if (!ok) ++failed;
    }
    if (failed) atomicAdd(n_failed, failed);
}


// =============================================================================================
// Bucketed multinomial resampling (device RNG).
//
// The direct kernel above does one 23-level binary search of the 80 MB CDF per output particle:
// ~1.2e8 scattered line requests at N = 1e7 (measured 1.4 ms, 83 % of GPU time in round-1
// profile a).  Output particles are exchangeable, so instead:
//   A  chunk counts      how many outputs descend from each CDF CHUNK (4096 source particles): exact
//                        Multinomial(N; W_chunk) counts.  k_bucket_counts: independent Poisson
//                        draws per chunk plus a short categorical top-up (see "Poissonisation" below);
//                        k_bucket_count / k_bucket_reduce (QSMC_COUNT_BY_DRAWS=1, the first implementation):
//                        every output draws u_i and is binned against the chunk edges in LDS;
//   B  (k_bucket_counts, last step) / k_bucket_plan   exclusive scans: first output slot of each
//                        chunk and a work list that splits heavy chunks into <= BUCKET_CAP outputs;
//   C  k_bucket_sample   one workgroup per work item scans ITS chunk of the weights into LDS (32 KB of CDF),
//                        draws the within-chunk position from an independent Philox word (given
//                        the counts, positions are i.i.d. uniform inside the chunk -- exact),
//                        searches in LDS, gathers x from the chunk's 32 KB window, kicks, checks
//                        validity and writes its outputs to consecutive slots.
// HBM traffic becomes streaming (read w + x once, write x' once); all scattered probes hit LDS.
// A postselection retry needs a fresh GLOBAL ancestor: that rare path falls back to the global
// search (same semantics as k_resample_philox: redraw ancestor and kick).
// =============================================================================================
constexpr int BUCKET_CHUNK = SCAN_CHUNK;            // 4096 source particles per bucket
constexpr int BUCKET_CAP = 2 * BUCKET_CHUNK;        // outputs per work item
constexpr int BUCKET_MAX_CHUNKS = 8192;             // skewed edges (68 KB) + counters (32 KB) + guide (16 KB) of LDS
constexpr int BUCKET_COUNT_BLOCKS = 256;            // one resident workgroup per CU
constexpr int BUCKET_COUNT_THREADS = 1024;

// Lower CDF edge of chunk c == upper edge of chunk c-1 == offsets[c] (k_scan_sums output; the chunk
// scan defines the last CDF entry of every chunk as exactly this number).
__device__ __forceinline__ double chunk_edge(const double *__restrict__ offsets, int64_t c) {
    return c <= 0 ? 0.0 : offsets[c];
}

// number of entries of a[0..m) that are <= u   (a in LDS or global)
__device__ __forceinline__ int upper_bound_i32(const double *a, int m, double u) {
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// LDS index skew: binary-search midpoints of a power-of-two table are multiples of 2048, 1024, ...
// elements, i.e. ONE bank for every lane (measured: 88 % of the sample kernel's LDS cycles were bank
// conflicts).  j + (j >> 5) + (j >> 10) sends those strides to distinct banks.
__device__ __forceinline__ int lds_skew(int j) { return j + (j >> 5) + (j >> 10); }
constexpr int BUCKET_CHUNK_LDS = BUCKET_CHUNK + (BUCKET_CHUNK >> 5) + (BUCKET_CHUNK >> 10) + 4;

// number of entries of the SKEWED LDS table a[skew(0..m)) that are <= u
__device__ __forceinline__ int upper_bound_skew(const double *a, int m, double u) {
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[lds_skew(mid)] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------
// Guide table: a 12-probe binary search of a 4096-entry LDS table costs ~12 instructions per probe.
// A uniform grid of cells over the table's value range, G[k] = #{entries whose cell < k}, brackets the
// answer for a query in cell k inside [G[k], G[k+1]] -- exactly, because guide_cell() is monotone and is
// applied identically to the entries and to the query -- typically 1-2 entries -> ~1 probe.
// k_bucket_count builds its table (cells over the chunk edges) with an LDS histogram + scan
// (build_guide); k_bucket_sample fills its table while it stores the scanned CDF (StoreLdsGuide).
// Exactness is unaffected either way: the answer always comes from comparing the entries with u.
// ---------------------------------------------------------------------------------------------
constexpr int GUIDE_BINS = 4096;              // k_bucket_count: cells over [0, 1) for the chunk edges
constexpr int SGUIDE_BINS = 2048;             // k_bucket_sample: cells over one chunk's 4096 CDF entries

template <int BINS>
__device__ __forceinline__ int guide_cell(double v, double lo, double scale) {
    const double t = (v - lo) * scale;
    int k = t > 0.0 ? (t < (double)(BINS - 1) ? (int)t : BINS - 1) : 0;
    return k;
}

// a: skewed LDS table of m non-decreasing values; G: int[GUIDE_BINS + 1] LDS; wtot: int[32] LDS.
template <int BT>
__device__ __forceinline__ void build_guide(const double *a, int m, double lo, double scale, int *G, int *wtot) {
    constexpr int PER = GUIDE_BINS / BT;
    static_assert(GUIDE_BINS % BT == 0, "GUIDE_BINS must be a multiple of the workgroup size");
    for (int k = threadIdx.x; k <= GUIDE_BINS; k += BT) G[k] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += BT) atomicAdd(&G[guide_cell<GUIDE_BINS>(a[lds_skew(j)], lo, scale) + 1], 1);
    __syncthreads();
    // inclusive scan of G[1..GUIDE_BINS]: thread owns PER consecutive cells
    int loc[PER];
    int run = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        run += G[1 + threadIdx.x * PER + q];
        loc[q] = run;
    }
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    int inc = run;
#pragma unroll
    for (int off = 1; off < QSMC_WAVE; off <<= 1) {
        const int t = __shfl_up(inc, off, QSMC_WAVE);
        if (lane >= off) inc += t;
    }
    if (lane == QSMC_WAVE - 1) wtot[wave] = inc;
    __syncthreads();
    int off0 = inc - run;
    for (int wv = 0; wv < wave; ++wv) off0 += wtot[wv];
#pragma unroll
    for (int q = 0; q < PER; ++q) G[1 + threadIdx.x * PER + q] = off0 + loc[q];
    __syncthreads();
}

// number of entries of the skewed table a[0..m) that are <= u, u lying in guide cell k
// guide_cell is monotone and is applied identically to the entries and to u, so entries in cells
// below k are <= u and entries in cells above k are > u: the answer lies in [G[k], G[k + 1]] exactly.
template <class GT>
__device__ __forceinline__ int guided_upper_bound(const double *a, int m, const GT *G, int k, double u) {
    int lo = G[k];
    int hi = G[k + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[lds_skew(mid)] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Philox stream layout of the bucketed resampler (round 0), two outputs per Philox block:
//   slot 0: block (c | t << 32)                   attempt t of chunk c's Poisson count (k_bucket_counts)
//           [QSMC_COUNT_BY_DRAWS: block (i >> 1), word (i & 1) = chunk draw of output i (k_bucket_count)]
//   slot 3: block (j >> 1), word (j & 1)          top-up draw j;  slot 4: block (i), word 0: removal i
//   slot 1: block (o >> 1), word (o & 1)          within-chunk position of slot o (k_bucket_sample)
//   slot 2: block (n >> 1), Box-Muller comp (n&1) n = o * d + q, q-th normal of slot o
// retries (round r >= 1) are per output: block (o, r, 0).u0 = global ancestor, (o, r, 1 + q/2) normals.
__global__ __launch_bounds__(BUCKET_COUNT_THREADS) void k_bucket_count(
    const double *__restrict__ offsets, int chunks, int64_t n_out, uint32_t k0, uint32_t k1,
    uint32_t epoch, unsigned int *__restrict__ hist /* [gridDim.x][chunks] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *edges = reinterpret_cast<double *>(smem);                       // skewed: upper edge of chunk c
    const int edges_len = lds_skew(chunks) + 4;
    unsigned int *cnt = reinterpret_cast<unsigned int *>(edges + edges_len);
    int *G = reinterpret_cast<int *>(cnt + chunks);
    int *wtot = G + GUIDE_BINS + 1;
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
        edges[lds_skew(c)] = chunk_edge(offsets, (int64_t)c + 1);
        cnt[c] = 0u;
    }
    __syncthreads();
    build_guide<BUCKET_COUNT_THREADS>(edges, chunks, 0.0, (double)GUIDE_BINS, G, wtot);
    const int64_t n_pairs = (n_out + 1) >> 1;
    for (int64_t pr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pr < n_pairs;
         pr += (int64_t)gridDim.x * blockDim.x) {
        PhiloxStream rng{(uint64_t)pr, (epoch << 16), k0, k1};
        double u[2];
        rng.uniforms(0, u[0], u[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (2 * pr + e < n_out) {
                // #edges <= u == chunk index; u in [0, 1) so its guide cell is exactly floor(u * 4096)
                int c = guided_upper_bound(edges, chunks, G, (int)(u[e] * (double)GUIDE_BINS), u[e]);
                if (c > chunks - 1) c = chunks - 1;              // u beyond cdf[n-1] (rounding): Q2 clamp
                atomicAdd(&cnt[c], 1u);
            }
        }
    }
    __syncthreads();
    unsigned int *row = hist + (size_t)blockIdx.x * chunks;
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) row[c] = cnt[c];
}

// ---------------------------------------------------------------------------------------------
// Chunk counts without drawing one uniform per output (k_bucket_count + k_bucket_reduce: 32 us at N = 1e7).
// Poissonisation: if T ~ Poisson(lambda) items are dealt to the chunks with probabilities p_c, the chunk
// counts are INDEPENDENT Poisson(lambda p_c) -- one draw per chunk, all in parallel, no tree and no depth --
// and given T they are Multinomial(T; p).  With lambda = n_out - kappa sqrt(n_out) (kappa = 5) T falls short of
// n_out by ~kappa sqrt(n_out) outputs, which are added as ordinary categorical draws (one uniform each,
// searched against the chunk edges: 1.6e4 draws instead of 1e7); Multinomial(T) + Multinomial(n_out - T) =
// Multinomial(n_out), exactly the law k_bucket_count samples.  Should T exceed n_out (probability 3e-7 per
// resample) the surplus is taken away again by removing T - n_out of the dealt items uniformly at random,
// which leaves an i.i.d. sample of size n_out: exact as well.
//
// poisson_draw: X ~ Poisson(mu), exact.  mu < 10: sequential search of the cdf from X = 0; otherwise PTRS
// (W. Hoermann, "The transformed rejection method for generating Poisson random variables", Insurance:
// Mathematics and Economics 12 (1993) 39): a squeeze accepts ~87 % of the proposals after one division and a
// floor.  Attempt t of chunk c takes its uniforms from Philox block (c | t << 32, round 0, slot 0) and the
// FIRST accepted attempt is the draw; G adjacent lanes evaluate attempts t0 .. t0 + G - 1 of one chunk at
// once and the lowest accepted one is taken (ballot + shuffle) -- the value a sequential loop returns, which
// is how the oracle's NumPy twin (oracle/philox.py: poisson_draw) computes it.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double stirling_tail(double k) {         // ln k! - [(k + 1/2) ln(k + 1) - (k + 1) + ln(2 pi) / 2]
    static constexpr double small[10] = {
        0.08106146679532726,  0.0413406959554093,   0.02767792568499834,  0.020790672103765093, 0.016644691189821193,
        0.013876128823070748, 0.01189670994589177,  0.010411265261972096, 0.009255462182712733, 0.00833056343336287};
    if (k < 10.0) return small[(int)k];
    const double rx = 1.0 / (k + 1.0), r2 = rx * rx;
    return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0 - (1.0 / 1680.0 - 1.0 / 1188.0 * r2) * r2) * r2) * r2) * rx;
}

// Called by whole waves.  Lanes [gbase, gbase + G) of a wave form the group of one chunk (same mu, node, active);
// G is a power of two <= 64.  Returns the draw to every lane of the group.
__device__ unsigned int poisson_draw(bool active, double mu, uint32_t node, uint32_t epoch_round, uint32_t k0,
                                     uint32_t k1, int G, int gbase) {
    const bool need = active && mu > 0.0;
    double y = 0.0;
    const bool by_search = need && mu < 10.0;
    if (by_search) {                                             // every lane of the group: same inputs, same value
        PhiloxStream rng{(uint64_t)node, epoch_round, k0, k1};
        double U, unused;
        rng.uniforms(0, U, unused);
        double pk = exp(-mu), cdf = pk, X = 0.0;
        while (U > cdf && X < 200.0) {
            X += 1.0;
            pk = pk * mu / X;
            cdf += pk;
        }
        y = X;
    }
    bool pending = need && !by_search;
    const double smu = sqrt(mu), lmu = log(mu);
    const double b = 0.931 + 2.53 * smu, a = -0.059 + 0.02483 * b;
    const double linva = log(1.1239 + 1.1328 / (b - 3.4)), vr = 0.9277 - 3.6224 / (b - 2.0);
    const int gl = (int)(threadIdx.x & (QSMC_WAVE - 1)) - gbase;
    const unsigned long long gmask = G >= 64 ? ~0ull : ((1ull << G) - 1ull);
    for (uint32_t t0 = 0; t0 < 4096u; t0 += (uint32_t)G) {
        if (__ballot(pending) == 0ull) break;                    // wave-uniform
        bool acc = false;
        double kk = 0.0;
        if (pending) {
            PhiloxStream rng{(uint64_t)node | ((uint64_t)(t0 + (uint32_t)gl) << 32), epoch_round, k0, k1};
            double U, V;
            rng.uniforms(0, U, V);
            const double u = U - 0.5, us = 0.5 - fabs(u);
            kk = floor((2.0 * a / us + b) * u + mu + 0.43);
            if (us >= 0.07 && V <= vr) acc = true;               // the squeeze
            else if (kk >= 0.0 && !(us < 0.013 && V > us)) {
                const double lhs = log(V) + linva - log(a / (us * us) + b);
                const double lgk = (kk + 0.5) * log(kk + 1.0) - (kk + 1.0) + 0.91893853320467274178 + stirling_tail(kk);
                acc = lhs <= -mu + kk * lmu - lgk;
            }
        }
        const unsigned long long grp = (__ballot(acc) >> gbase) & gmask;
        const int src = gbase + (grp ? __builtin_ctzll(grp) : 0);
        const double first = __shfl(kk, src, QSMC_WAVE);
        if (pending && grp) {
            y = first;
            pending = false;
        }
    }
    return need ? (unsigned int)y : 0u;
}

// single workgroup (1024 threads): slot_off[c] = exclusive scan of counts; item_off[c] = exclusive scan of
// ceil(counts / BUCKET_CAP); slot_off[chunks] = n_out, item_off[chunks] = #work items.  counts: global or LDS.
__device__ __forceinline__ void bucket_plan_block(const unsigned int *counts, int chunks,
                                                  long long *__restrict__ slot_off, int *__restrict__ item_off,
                                                  int *__restrict__ item_chunk) {
    __shared__ long long wtot_s[1024 / QSMC_WAVE];
    __shared__ int wtot_i[1024 / QSMC_WAVE];
    const int per = (chunks + 1023) / 1024;
    const int c0 = threadIdx.x * per, c1 = min(chunks, c0 + per);
    long long s = 0;
    int it = 0;
    for (int c = c0; c < c1; ++c) {
        s += counts[c];
        it += (int)((counts[c] + BUCKET_CAP - 1) / BUCKET_CAP);
    }
    // exclusive scan of the 1024 per-thread totals (integers: exact): shuffles inside a wave, 16 wave totals in LDS
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    long long inc_s = s;
    int inc_i = it;
#pragma unroll
    for (int off = 1; off < QSMC_WAVE; off <<= 1) {
        const long long ts = __shfl_up(inc_s, off, QSMC_WAVE);
        const int ti = __shfl_up(inc_i, off, QSMC_WAVE);
        if (lane >= off) {
            inc_s += ts;
            inc_i += ti;
        }
    }
    if (lane == QSMC_WAVE - 1) {
        wtot_s[wave] = inc_s;
        wtot_i[wave] = inc_i;
    }
    __syncthreads();
    long long so = inc_s - s;
    int io = inc_i - it;
    for (int wv = 0; wv < wave; ++wv) {
        so += wtot_s[wv];
        io += wtot_i[wv];
    }
    for (int c = c0; c < c1; ++c) {
        slot_off[c] = so;
        item_off[c] = io;
        const int items = (int)((counts[c] + BUCKET_CAP - 1) / BUCKET_CAP);
        for (int k = 0; k < items; ++k) item_chunk[io + k] = c;     // work item -> chunk map
        so += counts[c];
        io += items;
    }
    if (threadIdx.x == 1023) {                                   // (after the loop: so / io have run through its chunks)
        slot_off[chunks] = so;
        item_off[chunks] = io;
    }
}

// Barrier across the workgroups of ONE launch whose grid is small enough to be resident at once (16 here).  The
// arrival counter only ever grows: the host hands every launch the value all workgroups will have brought it to
// at each of its barriers, so nothing is reset.  No cache maintenance: the XCDs' L2s are not coherent with each
// other inside a launch, and a release / acquire fence pair at agent scope (L2 write-back + invalidate) measured
// ~5 us per barrier -- instead every word that crosses workgroups is written and read with agent-scope atomics
// (which go to the coherence point), and __syncthreads() has waited for this workgroup's own before the
// arrival is posted.  A bounded spin (~1 s): a launch that cannot become resident aborts (the next HIP call
// reports it) rather than hanging the queue or carrying on with half the data.
__device__ __forceinline__ void grid_barrier(unsigned long long *bar, unsigned long long target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) {                          // cannot happen with a resident grid: fail loudly
                __hip_atomic_fetch_add(bar + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_trap();
            }
        }
    }
    __syncthreads();
}
// A barrier with the cache maintenance, for bulk data written with plain stores (the redraw kernel's CDF): release
// (L2 write-back) before the arrival, acquire (invalidate) after the wait; ~5 us, on a path most resamples skip.
// Self-resetting (bar[0] arrivals, bar[1] departures: the last workgroup to leave clears both -- nobody can still
// be waiting then), so a launch that never reaches the barrier touches nothing.
__device__ __forceinline__ void grid_barrier_fenced(unsigned long long *bar) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 20)) __builtin_trap();          // cannot happen with a resident grid: fail loudly
        }
        if (__hip_atomic_fetch_add(bar + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1ull) {
            __hip_atomic_store(bar + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bar, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();                 // acquire for the whole workgroup: the CU's L1 and the XCD's L2 are shared
    }
    __syncthreads();
}
__device__ __forceinline__ void put_shared(unsigned int *p, unsigned int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int get_shared(const unsigned int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The chunk counts in one launch (each of the three steps alone is a ~5 us launch: the floor of a dependent
// kernel on this part):
//   0  (if the update kernel left tile sums) the chunk edges: see below;
//   1  counts[c] ~ Poisson(lambda mass_c / total), four lanes per chunk, chunks dealt to the workgroups;
//   2  every workgroup sums the counts to T (a few thousand integers), and takes its share of the n_out - T
//      categorical top-up draws against the chunk edges in LDS (1.6e4 draws on one CU were 25 us; spread over
//      16 they are 2), collected in an LDS histogram and added to extra[];  top-up draw j takes word (j & 1) of
//      Philox block (j >> 1, round 0, slot 3);
//   3  workgroup 0: counts += extra; should the Poisson total have overshot, thread 0 removes the surplus item
//      by item (removal i: word 0 of block (i, round 0, slot 4)); then the plan.
constexpr int POISSON_G = 4;
constexpr int BUCKET_COUNTS_BLOCKS = 16, BUCKET_COUNTS_THREADS = 1024;
__global__ __launch_bounds__(BUCKET_COUNTS_THREADS) void k_bucket_counts(
    double *offsets, TileSrc ts, unsigned long long *__restrict__ zero2, int chunks, int64_t n_out, double lambda,
    uint32_t k0, uint32_t k1, uint32_t epoch, unsigned int *counts, unsigned int *extra,
    long long *__restrict__ slot_off, int *__restrict__ item_off, int *__restrict__ item_chunk,
    unsigned long long *bar, unsigned long long bar_base) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *edges = reinterpret_cast<double *>(smem);
    unsigned int *hist = reinterpret_cast<unsigned int *>(edges + lds_skew(chunks) + 4);   // this workgroup's draws per chunk
    __shared__ unsigned long long total_s;
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    // ---- 0: the chunk edges.  Given the update kernel's tile sums (ts.tiles), EVERY workgroup forms the monotone
    // prefix of the chunk sums itself, straight into its LDS -- the same code on the same numbers in the same
    // order, so all agree bit for bit, and the separate one-workgroup scan launch (9 us) is gone; workgroup 0
    // also stores offsets[] for the sampler and clears the failed / retry counters.  Otherwise offsets[] is ready.
    static_assert(BUCKET_COUNTS_THREADS == SCAN_SUMS_THREADS, "scan_sums_block runs on this workgroup");
    if (ts.tiles) {
        const bool writer = blockIdx.x == 0;
        if (writer && threadIdx.x < 2) zero2[threadIdx.x] = 0ull;
        scan_sums_block(nullptr, (int64_t)chunks, ts, [&](int64_t i, double v) {
            if (i > 0) edges[lds_skew((int)i - 1)] = v;             // upper edge of chunk i - 1
            if (writer) offsets[i] = v;
        });
    } else {
        for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS) edges[lds_skew(c)] = offsets[c + 1];
    }
    for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS) hist[c] = 0u;
    if (threadIdx.x == 0) total_s = 0ull;
    __syncthreads();
    // ---- 1: Poisson counts ----
    const double total = edges[lds_skew(chunks - 1)];
    constexpr int PER_PASS = BUCKET_COUNTS_THREADS / POISSON_G;
    for (int c0 = (int)blockIdx.x * PER_PASS; c0 < chunks; c0 += (int)gridDim.x * PER_PASS) {   // (uniform per workgroup)
        const int c = c0 + (int)threadIdx.x / POISSON_G;
        const bool active = c < chunks;
        double mu = 0.0;
        if (active) {
            const double mass = edges[lds_skew(c)] - (c > 0 ? edges[lds_skew(c - 1)] : 0.0);
            mu = (mass > 0.0 && total > 0.0) ? lambda * mass / total : 0.0;
        }
        const unsigned int x = poisson_draw(active, mu, (uint32_t)c, (epoch << 16), k0, k1, POISSON_G, lane & ~(POISSON_G - 1));
        if (active && (lane & (POISSON_G - 1)) == 0) {
            put_shared(&counts[c], x);
            put_shared(&extra[c], 0u);
        }
    }
    grid_barrier(bar, bar_base + gridDim.x);
    // ---- 2: the total, and this workgroup's share of the top-up ----
    unsigned long long mine = 0ull;
    for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS)
        mine += get_shared(&counts[c]);
    for (int off = QSMC_WAVE / 2; off > 0; off >>= 1) mine += __shfl_down(mine, off, QSMC_WAVE);
    if (lane == 0 && mine) atomicAdd(&total_s, mine);
    __syncthreads();
    const long long T = (long long)total_s;
    if (T < n_out) {
        const int64_t deficit = n_out - T, n_pairs = (deficit + 1) >> 1;
        for (int64_t pr = (int64_t)blockIdx.x * BUCKET_COUNTS_THREADS + threadIdx.x; pr < n_pairs;
             pr += (int64_t)gridDim.x * BUCKET_COUNTS_THREADS) {
            PhiloxStream rng{(uint64_t)pr, (epoch << 16), k0, k1};
            double u[2];
            rng.uniforms(3, u[0], u[1]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (2 * pr + e < deficit) {
                    int c = upper_bound_skew(edges, chunks, u[e]);     // #edges <= u == chunk index
                    if (c > chunks - 1) c = chunks - 1;                // u beyond cdf[n-1] (rounding): Q2 clamp
                    atomicAdd(&hist[c], 1u);
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS)
            if (hist[c]) atomicAdd(&extra[c], hist[c]);
    }
    grid_barrier(bar, bar_base + 2ull * gridDim.x);
    if (blockIdx.x != 0) return;
    // ---- 3: final counts and the plan ----
    unsigned int *cnt = hist;
    for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS)
        cnt[c] = get_shared(&counts[c]) + get_shared(&extra[c]);
    __syncthreads();
    if (T > n_out) {                                             // (uniform branch; ~3e-7 of the resamples)
        if (threadIdx.x == 0) {
            long long left = T;
            for (long long i = 0; i < T - n_out; ++i, --left) {
                PhiloxStream rng{(uint64_t)i, (epoch << 16), k0, k1};
                double u, unused;
                rng.uniforms(4, u, unused);
                long long target = (long long)(u * (double)left);   // which of the remaining items goes
                if (target > left - 1) target = left - 1;
                int c = 0;
                for (long long run = (long long)cnt[0]; run <= target; run += (long long)cnt[c]) ++c;
                cnt[c] -= 1u;
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS) counts[c] = cnt[c];
    bucket_plan_block(cnt, chunks, slot_off, item_off, item_chunk);
}

// counts[c] = sum_g hist[g][c].  A workgroup takes 64 chunks; its four waves each sum a quarter of the rows
// (coalesced 256-byte row segments), LDS combines them: 4x the workgroups and a quarter of the dependent
// loads per thread of the one-thread-per-chunk version (8.8 -> ~4 us, the launch floor).
__global__ __launch_bounds__(QSMC_BLOCK) void k_bucket_reduce(const unsigned int *__restrict__ hist, int rows,
                                                              int chunks, unsigned int *__restrict__ counts) {
    __shared__ unsigned int part[QSMC_WAVES_PER_BLOCK][QSMC_WAVE];
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
    const int c = blockIdx.x * QSMC_WAVE + lane;
    unsigned int s = 0;
    if (c < chunks) {
#pragma unroll 16
        for (int g = wave; g < rows; g += QSMC_WAVES_PER_BLOCK) s += hist[(size_t)g * chunks + c];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < chunks) {
#pragma unroll
        for (int wv = 1; wv < QSMC_WAVES_PER_BLOCK; ++wv) s += part[wv][lane];
        counts[c] = s;
    }
}

__global__ __launch_bounds__(1024) void k_bucket_plan(const unsigned int *__restrict__ counts, int chunks,
                                                      long long *__restrict__ slot_off,
                                                      int *__restrict__ item_off,
                                                      int *__restrict__ item_chunk) {
    bucket_plan_block(counts, chunks, slot_off, item_off, item_chunk);
}

constexpr int BUCKET_RLIST_CAP = 1024;               // per-workgroup list of outputs that need a global redraw

// Draw + kick of one redraw round (round >= 1) of output slot o from the GLOBAL CDF.
template <int DM>
__device__ __forceinline__ bool redraw_rounds(int kind, int d, double min_freq, const double *__restrict__ x_in,
                                              int64_t ldx_in, int64_t n_in, const double *__restrict__ cdf,
                                              const LWArgs &lw, uint32_t k0, uint32_t k1, uint32_t epoch,
                                              int maxiter, int64_t o, double *p) {
    for (int round = 1; round < maxiter; ++round) {
        PhiloxStream rng{(uint64_t)o, (epoch << 16) | (uint32_t)round, k0, k1};
        double u0, unused;
        rng.uniforms(0, u0, unused);
        const int64_t j = search_right(cdf, n_in, u0);
        double zz[DM];
#pragma unroll
        for (int q = 0; q < DM; q += 2) {
            if (q < d) {
                double z0, z1;
                rng.normals(1 + (q >> 1), z0, z1);
                zz[q] = z0;
                if (q + 1 < DM) zz[q + 1] = z1;
            }
        }
#pragma unroll
        for (int m = 0; m < DM; ++m) {
            if (m < d) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < DM; ++q)
                    if (q < d) s += lw.S[m * d + q] * zz[q];
                p[m] = (lw.a * x_in[m * ldx_in + j] + (1.0 - lw.a) * lw.mean[m]) + s;
            }
        }
        if (model_valid(kind, p, min_freq)) return true;
    }
    return false;
}

// chunk_scan_block sink of the sampler: the entry goes to the skewed LDS table and, in the same pass,
// the guide table is filled -- entry j is the first one whose cell is >= k for every cell k in
// (cell(prev), cell(v)], so G[k] = j there (G[0] = 0, G[SGUIDE_BINS] = len): no histogram, no atomics,
// no extra barrier.  Runs longer than 8 cells (a dominant weight) are filled by the whole wave.
struct StoreLdsGuide {
    double *lcdf;
    unsigned short *G;
    double lo_edge, gscale;
    bool use_guide;                                // workgroup-uniform
    int len;
    int cp;                                        // cell of the previous entry (carried along the lane's run)
    __device__ __forceinline__ void operator()(int j, double v, double prev, bool live) {
        if (live) lcdf[lds_skew(j)] = v;
        if (!use_guide) return;
        const int lane = threadIdx.x & (QSMC_WAVE - 1);
        if ((j & (SCAN_PER_LANE - 1)) == 0) cp = j == 0 ? -1 : guide_cell<SGUIDE_BINS>(prev, lo_edge, gscale);
        const int cj = live ? guide_cell<SGUIDE_BINS>(v, lo_edge, gscale) : cp;
        const unsigned short js = (unsigned short)j;
        // straight-line for the common run lengths 0..4
        if (cj > cp) G[cp + 1] = js;
        if (cj > cp + 1) G[cp + 2] = js;
        if (cj > cp + 2) G[cp + 3] = js;
        if (cj > cp + 3) G[cp + 4] = js;
        if (live && j == len - 1) G[SGUIDE_BINS] = (unsigned short)len;
        unsigned long long long_runs = __ballot(cj > cp + 4);
        while (long_runs) {                        // a dominant weight: the wave fills the run together
            const int src = __ffsll((long long)long_runs) - 1;
            long_runs &= long_runs - 1;
            const int s0 = __shfl(cp + 5, src, QSMC_WAVE), e0 = __shfl(cj, src, QSMC_WAVE);
            const int jj = __shfl(j, src, QSMC_WAVE);
            for (int k = s0 + lane; k <= e0; k += QSMC_WAVE) G[k] = (unsigned short)jj;
        }
        cp = cj;
    }
};

// One workgroup per work item.  The chunk's CDF is SCANNED HERE from the weights (bit-identical
// to k_chunk_scan), so the CDF never touches HBM; a particle that fails postselection on its first
// try is queued for k_bucket_retry, which alone needs the (then materialised) global CDF.
// Occupancy: 44 KB of LDS allows three workgroups per CU; the small-d instantiations are held to 80
// VGPRs (6 waves/SIMD) so that the third one fits -- the kernel is VALU-issue bound and the extra
// waves hide the LDS search and gather latency (121 -> 110 us at N = 1e7, d = 1).
template <int D, int BT>   // D = 0: runtime d; BT = threads per workgroup
__attribute__((amdgpu_waves_per_eu(D >= 1 && D <= 2 ? 6 : 1, 8)))
__global__ __launch_bounds__(BT) void k_bucket_sample(
    int kind, int d_rt, double min_freq, int postselect, const double *__restrict__ x_in, int64_t ldx_in,
    int64_t n_in, const double *__restrict__ w, double inv_norm, const double *__restrict__ offsets,
    int chunks, const long long *__restrict__ slot_off,
    const int *__restrict__ item_off, const int *__restrict__ item_chunk, LWArgs lw, uint32_t k0, uint32_t k1,
    uint32_t epoch, int maxiter, double *__restrict__ x_out, OutPlace pl,
    unsigned long long *__restrict__ n_failed, unsigned int *__restrict__ retry_list,
    unsigned long long *__restrict__ retry_count) {
    constexpr int DM = D > 0 ? D : QSMC_MAX_D;
    const int d = D > 0 ? D : d_rt;
    __shared__ __attribute__((aligned(16))) double lcdf[BUCKET_CHUNK_LDS];
    __shared__ unsigned short lguide[SGUIDE_BINS + 2];
    __shared__ double wave_tot[SCAN_WAVES];
    __shared__ unsigned short rlist[BUCKET_RLIST_CAP];          // slot - o_begin < BUCKET_CAP
    static_assert(BT >= SCAN_THREADS, "the in-sampler chunk scan needs 512 threads");
    __shared__ int rcount;
    __shared__ unsigned long long rbase;
    if ((int)blockIdx.x >= item_off[chunks]) return;
    const int c = item_chunk[blockIdx.x];
    const int part = (int)blockIdx.x - item_off[c];
    const long long slot0 = slot_off[c], n_c = slot_off[c + 1] - slot0;
    const long long t0 = (long long)part * BUCKET_CAP;
    const long long t1 = t0 + BUCKET_CAP < n_c ? t0 + BUCKET_CAP : n_c;
    const int64_t base = (int64_t)c * BUCKET_CHUNK;
    const int len = (int)((n_in - base) < BUCKET_CHUNK ? (n_in - base) : BUCKET_CHUNK);
    if (threadIdx.x == 0) rcount = 0;
    const double lo_edge = chunk_edge(offsets, c);
    const double hi_edge = offsets[c + 1];
    const double gscale = (double)SGUIDE_BINS / (hi_edge - lo_edge);
    const bool use_guide = hi_edge > lo_edge && gscale < 1e300;     // workgroup-uniform
    chunk_scan_block(w, n_in, inv_norm, offsets, (int64_t)c, wave_tot,
                     StoreLdsGuide{lcdf, lguide, lo_edge, gscale, use_guide, len, -1});
    __syncthreads();
    const int64_t o_begin = slot0 + t0, o_end = slot0 + t1;         // this item's output slots
    unsigned long long failed = 0;
    // pairs of output slots (2P, 2P+1) share their Philox blocks; a pair straddling two work items is
    // evaluated by both, each writing only its own half.
    // Stage A (ancestors): position Philox -> guided LDS search -> gather of x (d <= 4: into registers).
    // Stage B (kick): normals Philox + Box-Muller -> Liu-West combine -> validity -> store.
    // A is issued first so that its LDS round trips and the L2/HBM gather are in flight during B's ~600
    // cycles of independent arithmetic (110 -> 100 us at N = 1e7; issuing A of the NEXT pair ahead of B --
    // a software pipeline -- was measured too and is slower, 115 us: spills).  Both halves of a pair are
    // searched even if one belongs to the neighbouring work item: no divergence, cheap.
    constexpr bool EARLY = DM <= 4;
    struct Anc {
        int jl[2];
        double xg[2][EARLY ? DM : 1];
    };
    auto stage_a = [&](int64_t P, Anc &an) {
        PhiloxStream rng{(uint64_t)P, (epoch << 16), k0, k1};
        double upos[2];
        rng.uniforms(1, upos[0], upos[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // position inside this chunk: given the counts, uniform on [lo_edge, hi_edge)
            const double u = lo_edge + upos[e] * (hi_edge - lo_edge);
            int j = use_guide ? guided_upper_bound(lcdf, len, lguide, guide_cell<SGUIDE_BINS>(u, lo_edge, gscale), u)
                              : upper_bound_skew(lcdf, len, u);
            an.jl[e] = j > len - 1 ? len - 1 : j;
            if (EARLY) {
#pragma unroll
                for (int m = 0; m < DM; ++m)
                    if (m < d) an.xg[e][m] = x_in[m * ldx_in + base + an.jl[e]];
            }
        }
    };
    auto stage_b = [&](int64_t P, const Anc &an) {
        double z[2 * DM];
        PhiloxStream nrm{0, (epoch << 16), k0, k1};
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            if (k < d) {
                nrm.particle = (uint64_t)P * (uint64_t)d + (uint64_t)k;
                nrm.normals(2, z[2 * k], z[2 * k + 1]);
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t o = 2 * P + e;
            if (o >= o_begin && o < o_end) {
                double p[DM];
#pragma unroll
                for (int m = 0; m < DM; ++m) {
                    if (m < d) {
                        double sm = 0.0;
#pragma unroll
                        for (int q = 0; q < DM; ++q)
                            if (q < d) sm += lw.S[m * d + q] * z[e * d + q];
                        const double xa = EARLY ? an.xg[e][m] : x_in[m * ldx_in + base + an.jl[e]];
                        p[m] = (lw.a * xa + (1.0 - lw.a) * lw.mean[m]) + sm;
                    }
                }
                bool ok = !postselect || model_valid(kind, p, min_freq);
                if (!ok && maxiter > 1) {
                    // queue for k_bucket_retry (needs the global CDF)
                    const int idx = atomicAdd(&rcount, 1);
                    if (idx < BUCKET_RLIST_CAP) rlist[idx] = (unsigned short)(o - o_begin);
                    else retry_list[atomicAdd(retry_count, 1ull)] = (unsigned int)o;   // rare overflow path
                    ok = true;                      // decided later
                }
                const int64_t row = place_row(pl, o);
#pragma unroll
                for (int m = 0; m < DM; ++m)
                    if (m < d) x_out[m * pl.ld_m + row * pl.ld_s] = p[m];
                if (!ok) ++failed;
            }
        }
    };
    for (int64_t P = (o_begin >> 1) + threadIdx.x; 2 * P < o_end; P += BT) {
        Anc an;
        stage_a(P, an);
        stage_b(P, an);
    }
    if (failed) atomicAdd(n_failed, failed);
    __syncthreads();
    const int nl = rcount < BUCKET_RLIST_CAP ? rcount : BUCKET_RLIST_CAP;
    if (nl == 0) return;
    if (threadIdx.x == 0) rbase = atomicAdd(retry_count, (unsigned long long)nl);   // one atomic per workgroup
    __syncthreads();
    for (int i = threadIdx.x; i < nl; i += BT) retry_list[rbase + i] = (unsigned int)(o_begin + rlist[i]);
}

// Second chance for the queued outputs: redraw ancestor and kick from the global CDF (rounds 1..).  One launch,
// resident grid: nothing queued (most resamples) -> leave at once (an empty launch is ~5 us;
// the former pair -- materialise the CDF behind a gate, then redraw -- was two of them).  Otherwise every
// workgroup scans its share of the chunks into the global CDF, all meet at a barrier, and the queue is worked off.
// Held to 128 VGPRs (4 waves per SIMD): two workgroups fit a CU, so 256 are resident on half the CUs and two
// processes sharing a GPU (as the tests do) both stay resident; the scan phase takes ~10 rounds instead of 19.
constexpr int REDRAW_BLOCKS = 256;
template <int DM>     // particle dimension bound: 4 (registers) or QSMC_MAX_D
__attribute__((amdgpu_waves_per_eu(4, 8)))
__global__ __launch_bounds__(SCAN_THREADS) void k_bucket_redraw(
    int kind, int d, double min_freq, const double *__restrict__ x_in, int64_t ldx_in, int64_t n_in,
    const double *__restrict__ w, double inv_norm, const double *__restrict__ offsets, int64_t chunks, double *cdf,
    LWArgs lw, uint32_t k0, uint32_t k1, uint32_t epoch, int maxiter, double *__restrict__ x_out, OutPlace pl,
    const unsigned int *__restrict__ retry_list, const unsigned long long *__restrict__ retry_count,
    unsigned long long *__restrict__ n_failed, unsigned long long *bar) {
    __shared__ double wave_tot[SCAN_WAVES];
    const unsigned long long cnt = *retry_count;
    if (cnt == 0ull) return;
    for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        chunk_scan_block(w, n_in, inv_norm, offsets, c, wave_tot, StoreGlobal{cdf + c * SCAN_CHUNK});
        __syncthreads();                                         // wave_tot is reused by the next chunk
    }
    grid_barrier_fenced(bar);
    unsigned long long failed = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * SCAN_THREADS + threadIdx.x; i < cnt;
         i += (unsigned long long)gridDim.x * SCAN_THREADS) {
        const int64_t o = (int64_t)retry_list[i];
        double p[DM];
        const bool ok = redraw_rounds<DM>(kind, d, min_freq, x_in, ldx_in, n_in, cdf, lw, k0, k1, epoch,
                                                  maxiter, o, p);
        const int64_t row = place_row(pl, o);      // like the in-thread loop: the last round's value stays
        for (int m = 0; m < d; ++m) x_out[m * pl.ld_m + row * pl.ld_s] = p[m];
        if (!ok) ++failed;
    }
    if (failed) atomicAdd(n_failed, failed);
}

// copies the failed-particle counter into pinned host memory (read later, after any stream sync)
__global__ void k_publish_counter(const unsigned long long *__restrict__ counter, double *__restrict__ mapped_slot) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *mapped_slot = (double)*counter;
}

__global__ __launch_bounds__(QSMC_BLOCK) void k_prior_uniform_philox(
    int kind, int d, double min_freq, int postselect, LWArgs box /* mean = lo, S[0..d) = hi - lo */,
    int64_t n, uint32_t k0, uint32_t k1, uint32_t epoch, int maxiter, double *__restrict__ x_out,
    int64_t ldx_out, unsigned long long *__restrict__ n_failed) {
    unsigned long long failed = 0;
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double p[QSMC_MAX_D];
        bool ok = false;
        for (int round = 0; round < maxiter && !ok; ++round) {
            PhiloxStream rng{(uint64_t)i, (epoch << 16) | (uint32_t)round, k0, k1};
            for (int q = 0; q < d; q += 2) {
                double u0, u1;
                rng.uniforms(q >> 1, u0, u1);
                p[q] = box.mean[q] + u0 * box.S[q];                         // lo + z * delta (:818-819)
                if (q + 1 < d) p[q + 1] = box.mean[q + 1] + u1 * box.S[q + 1];
            }
            ok = !postselect || model_valid(kind, p, min_freq);
        }
        for (int m = 0; m < d; ++m) x_out[m * ldx_out + i] = p[m];
        if (!ok) ++failed;
    }
    if (failed) atomicAdd(n_failed, failed);
}

// =============================================================================================
// time-step updates (smc.py:447-449, Model.update_timestep): the cloud takes a random-walk step between
// data.  x[m][i] += scale[m] * z, in place; rows with scale 0 do not move and are not touched.
//   z given (device, [row r of the walking parameters][i]): the host drew the steps (parity mode: the
//     reference's np.random.normal call, or an arbitrary step distribution of a RandomWalkModel);
//   z == nullptr: standard normals from Philox -- pair index P = i >> 1 shares a block across the two
//     particles of a pair for ONE walking parameter r: block (P, epoch, slot r), Box-Muller comp i & 1.
// HBM-bound: reads and writes the walking rows once (16 B per particle per walking parameter).
// =============================================================================================
struct WalkArgs {
    double scale[QSMC_MAX_D];
    int row[QSMC_MAX_D];        // parameter index of walking row r
    int n_rw;
};

__global__ __launch_bounds__(QSMC_BLOCK) void k_random_walk(double *__restrict__ x, int64_t ldx, int64_t n,
                                                            WalkArgs wa, const double *__restrict__ z, int64_t ldz,
                                                            uint32_t k0, uint32_t k1, uint32_t epoch) {
    const int64_t n_pairs = (n + 1) >> 1;
    for (int64_t P = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; P < n_pairs;
         P += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t i0 = 2 * P, i1 = 2 * P + 1;
        for (int r = 0; r < wa.n_rw; ++r) {
            double z0, z1;
            if (z) {
                z0 = z[r * ldz + i0];
                z1 = i1 < n ? z[r * ldz + i1] : 0.0;
            } else {
                PhiloxStream rng{(uint64_t)P, (epoch << 16), k0, k1};
                rng.normals((uint32_t)r, z0, z1);
            }
            double *row = x + (int64_t)wa.row[r] * ldx;
            row[i0] += wa.scale[r] * z0;
            if (i1 < n) row[i1] += wa.scale[r] * z1;
        }
    }
}

// =============================================================================================
// tomography canonicalize: per-particle dim x dim complex Hermitian Jacobi, clamp, re-expand
// =============================================================================================
template <int DIM>
__global__ __launch_bounds__(QSMC_BLOCK) void k_tomo_canon(const double *__restrict__ basis,
                                                           double *__restrict__ x, int64_t ldx, int64_t n,
                                                           int allow_subnormalized) {
    constexpr int D = DIM * DIM;
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double p[D];
#pragma unroll
        for (int a = 0; a < D; ++a) p[a] = x[a * ldx + i];
        if (tomo_canon_particle<DIM>(basis, p, allow_subnormalized != 0)) {
#pragma unroll
            for (int a = 0; a < D; ++a) x[a * ldx + i] = p[a];
        }
    }
}

// Two passes for dim = 4 (2 qubits): the eigendecomposition is ~6000 flops and 240 VGPRs per particle, but a
// particle whose rho is positive definite only needs its trace renormalised -- and about two thirds of a
// freshly resampled cloud are (36 % non-PSD measured after a Liu-West kick).  Deciding per lane inside one
// kernel would not help (a wave is as slow as its slowest lane), so pass 1 classifies with a pivot test
// (tomo_clearly_positive), finishes the clear cases and compacts the others into an index list
// (one atomic per wave); pass 2 runs the Jacobi path on the list only, densely packed.
template <int DIM>
__global__ __launch_bounds__(QSMC_BLOCK) void k_tomo_classify(const double *__restrict__ basis,
                                                              double *__restrict__ x, int64_t ldx, int64_t n,
                                                              int allow_subnormalized, unsigned int *__restrict__ list,
                                                              unsigned int *__restrict__ count) {
    constexpr int D = DIM * DIM;
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    for (int64_t i0 = (int64_t)blockIdx.x * QSMC_BLOCK; i0 < n; i0 += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t i = i0 + threadIdx.x;
        bool hard = false;
        if (i < n) {
            double p[D];
#pragma unroll
            for (int a = 0; a < D; ++a) p[a] = x[a * ldx + i];
            if (tomo_clearly_positive<DIM>(basis, p)) {
                if (!allow_subnormalized) {                   // tomography/models.py:194-209
                    const double nrm = p[0] * sqrt((double)DIM);
#pragma unroll
                    for (int a = 0; a < D; ++a) x[a * ldx + i] = p[a] / nrm;
                }
            } else {
                hard = true;
            }
        }
        const unsigned long long m = __ballot(hard);
        if (m) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(count, (unsigned int)__popcll(m));
            base = __shfl(base, 0, QSMC_WAVE);
            if (hard) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned int)i;
        }
    }
}

template <int DIM>
__global__ __launch_bounds__(QSMC_BLOCK) void k_tomo_canon_list(const double *__restrict__ basis,
                                                                double *__restrict__ x, int64_t ldx,
                                                                int allow_subnormalized,
                                                                const unsigned int *__restrict__ list,
                                                                const unsigned int *__restrict__ count) {
    constexpr int D = DIM * DIM;
    const unsigned int m = *count;
    for (unsigned int t = blockIdx.x * QSMC_BLOCK + threadIdx.x; t < m; t += gridDim.x * QSMC_BLOCK) {
        const int64_t i = list[t];
        double p[D];
#pragma unroll
        for (int a = 0; a < D; ++a) p[a] = x[a * ldx + i];
        if (tomo_canon_particle<DIM>(basis, p, allow_subnormalized != 0)) {
#pragma unroll
            for (int a = 0; a < D; ++a) x[a * ldx + i] = p[a];
        }
    }
}

// =============================================================================================
// host-side helpers
// =============================================================================================
static int make_exp_args(const qsmc_model_t *model, const qsmc_expparam_t *ep, int64_t outcome,
                         ExpArgs *out) {
    memset(out, 0, sizeof(*out));
    out->t = ep->t;
    out->w_ = ep->w_;
    out->n_meas = (double)ep->n_meas;
    out->m = (double)ep->m;
    out->reference = ep->reference;
    out->d = model->d;
    out->lik_pow = (model->likelihood_power == 1.0) ? 0.0 : model->likelihood_power;
    for (int i = 0; i < QSMC_MAX_D; ++i) out->meas[i] = ep->meas[i];
    out->comb = 1.0;
    out->log_comb = 0.0;
    if (model->kind == QSMC_MODEL_BINOMIAL_PRECESSION || model->kind == QSMC_MODEL_BINOMIAL_RB ||
        model->kind == QSMC_MODEL_BINOMIAL_RB_INTERLEAVED) {
        const double n = (double)ep->n_meas, k = (double)outcome;
        if (outcome >= 0 && k <= n) {
            out->log_comb = lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0);
            // exact integer binomial coefficient while it fits a double exactly-ish
            long double c = 1.0L;
            const int64_t kk = (outcome < (int64_t)(ep->n_meas - (uint64_t)outcome))
                                   ? outcome : (int64_t)(ep->n_meas - (uint64_t)outcome);
            bool finite = true;
            for (int64_t j = 1; j <= kk; ++j) {
                c = c * (long double)(ep->n_meas - (uint64_t)kk + (uint64_t)j) / (long double)j;
                if (c > 1.0e300L) { finite = false; break; }
            }
            out->comb = finite ? (double)c : INFINITY;
        }
    }
    return QSMC_OK;
}

static int check_model(const qsmc_model_t *m) {
    if (!m) return QSMC_ERR_INVALID;
    switch (m->kind) {
        case QSMC_MODEL_PRECESSION:
        case QSMC_MODEL_BINOMIAL_PRECESSION: return m->d == 1 ? QSMC_OK : QSMC_ERR_INVALID;
        case QSMC_MODEL_RB:
        case QSMC_MODEL_BINOMIAL_RB: return m->d == 3 ? QSMC_OK : QSMC_ERR_INVALID;
        case QSMC_MODEL_RB_INTERLEAVED:
        case QSMC_MODEL_BINOMIAL_RB_INTERLEAVED: return m->d == 4 ? QSMC_OK : QSMC_ERR_INVALID;
        case QSMC_MODEL_TOMOGRAPHY: return (m->d >= 1 && m->d <= QSMC_MAX_D) ? QSMC_OK : QSMC_ERR_INVALID;
        case QSMC_MODEL_UNKNOWN_T2: return m->d == 2 ? QSMC_OK : QSMC_ERR_INVALID;
        default: return QSMC_ERR_INVALID;
    }
}

static ReduceOut make_reduce(qsmc_ctx *h, bool want_host, double *stats4) {
    ReduceOut ro;
    ro.partials = h->partials;
    ro.out_dev = h->red_out;
    ro.out_mapped = want_host ? h->mapped_dev : nullptr;
    ro.stats4 = stats4;
    ro.flag = want_host ? h->flag_dev : nullptr;
    ro.seq = want_host ? ++h->seq : 0ull;
    ro.failed_src = reinterpret_cast<const unsigned long long *>(h->counter);
    ro.failed_dst = want_host ? h->mapped_dev + (REDUCE_OUT_MAX - 1) : nullptr;
    ro.tile_sums = nullptr;
    return ro;
}

// Wait for the reduction that was armed with the current h->seq.  hipStreamSynchronize costs ~12 us
// after an (already finished) kernel on this stack; spinning on a pinned word the reducing workgroup
// writes after a system-scope fence costs ~7 us (tools/lat/lat.hip).  Falls back to a real
// synchronise after 20 ms so that a faulted launch reports its HIP error instead of hanging.
static int wait_reduction(qsmc_ctx *h, hipStream_t s) {
    const unsigned long long want = h->seq;
    volatile unsigned long long *f = h->flag;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (*f == want) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return QSMC_OK;
        }
        __builtin_ia32_pause();
        if ((spins & 0xfff) == 0xfff &&
            std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    HIP_TRY(h, hipStreamSynchronize(s));
    return QSMC_OK;
}

static int launch_reduce(qsmc_ctx *h, int ns, int grid, const ReduceOut &ro, hipStream_t s) {
    switch (ns) {
#define LR(N)                                                                                   \
    case N:                                                                                     \
        hipLaunchKernelGGL((k_reduce_partials<N>), dim3(1), dim3(QSMC_BLOCK), 0, s, grid, ro);  \
        break;
        LR(3) LR(4) LR(5) LR(6) LR(8) LR(10) LR(12) LR(15) LR(16) LR(17) LR(20) LR(24) LR(26) LR(29) LR(32) LR(33)
        LR(38) LR(64) LR(80) LR(128)
#undef LR
        default: return QSMC_ERR_INVALID;
    }
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

// After a grid-reducing launch: synchronise and hand the totals (already in pinned memory) back.
// out layout [sum, sumsq, bad, extra..., min] with `ns` sums.
static int collect_stats(qsmc_ctx *h, int ns, qsmc_update_stats_t *stats_host, double *extra_host, int n_extra,
                         hipStream_t s) {
    if (!stats_host && !extra_host) return QSMC_OK;
    const int rc = wait_reduction(h, s);
    if (rc) return rc;
    if (stats_host) {
        stats_host->sum = h->mapped[0];
        stats_host->sumsq = h->mapped[1];
        stats_host->n_bad = h->mapped[2];
        stats_host->min = h->mapped[ns];
    }
    if (extra_host) memcpy(extra_host, h->mapped + 3, (size_t)n_extra * sizeof(double));
    return QSMC_OK;
}

// Next (start, stop) event pair of the profiling ring, or (null, null) when profiling is off.
static void prof_events(qsmc_ctx *h, int tag, hipEvent_t *e0, hipEvent_t *e1) {
    if (!h->profiling || !h->prof_ev) return;
    if (h->prof_seen[tag & 3]++ % (unsigned)h->prof_stride != 0) return;
    const int slot = h->prof_n % QSMC_PROF_CAP;         // a ring: beyond the capacity the oldest are overwritten
    *e0 = h->prof_ev[2 * slot];
    *e1 = h->prof_ev[2 * slot + 1];
    h->prof_tag[slot] = (unsigned char)tag;
    ++h->prof_n;
}

template <int KIND>
static void launch_update(qsmc_ctx *h, bool vec2, int grid, hipStream_t s, const double *x, int64_t ldx,
                          int64_t n, const double *w_in, double *w_out, double prev_norm, const ExpArgs &e,
                          int64_t outcome, const ReduceOut &ro) {
    // In profiling mode the launch carries start/stop events, so the elapsed time is the kernel's
    // own execution (what rocprofv3 --kernel-trace reports), not launch latency.
    hipEvent_t e0 = nullptr, e1 = nullptr;
    prof_events(h, w_in ? QSMC_PROF_UPDATE : QSMC_PROF_UPDATE_ONES, &e0, &e1);
#define LU(V, O)                                                                                          \
    do {                                                                                                  \
        if (e.lik_pow != 0.0)                                                                             \
            hipExtLaunchKernelGGL((k_update_fused<KIND, V, O, true>), dim3(grid), dim3(QSMC_BLOCK), 0, s, e0, e1, 0, \
                                  x, ldx, n, w_in, w_out, prev_norm, e, outcome, ro);                     \
        else                                                                                              \
            hipExtLaunchKernelGGL((k_update_fused<KIND, V, O, false>), dim3(grid), dim3(QSMC_BLOCK), 0, s, e0, e1, 0, \
                                  x, ldx, n, w_in, w_out, prev_norm, e, outcome, ro);                     \
    } while (0)
    if (vec2 && w_in) LU(2, false);
    else if (vec2) LU(2, true);
    else if (w_in) LU(1, false);
    else LU(1, true);
#undef LU
}

template <int MODE>
static int weights_pass(qsmc_ctx *h, const double *L, int64_t n, const double *w_in, double *w_out,
                        double norm, double *stats_dev, qsmc_update_stats_t *stats_host, hipStream_t s) {
    const int grid = grid_for(n, QSMC_BLOCK * 4);
    int rc = ensure_partials(h, (size_t)grid * 4);
    if (rc) return rc;
    const ReduceOut ro = make_reduce(h, stats_host != nullptr, stats_dev);
    hipLaunchKernelGGL((k_weights_pass<MODE>), dim3(grid), dim3(QSMC_BLOCK), 0, s, L, n, w_in, w_out, norm, ro);
    HIP_TRY(h, hipGetLastError());
    rc = launch_reduce(h, 3, grid, ro, s);
    if (rc) return rc;
    return collect_stats(h, 3, stats_host, nullptr, 0, s);
}

template <int KIND, int NO>
static int hyp_launch(qsmc_ctx *h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                      const double *w, double norm, const qsmc_expparam_t *exp, const int64_t *outcomes, int n_o,
                      const double *shift, double *out_host, hipStream_t s) {
    constexpr int D = Model<KIND>::D <= 4 ? Model<KIND>::D : 0;
    constexpr int PER = 2 + 2 * D;
    constexpr int NS = NO * PER;
    static_assert(NS + 1 <= REDUCE_OUT_MAX - 2, "reduce buffers too small");
    const int grid = grid_for(n, QSMC_BLOCK * 4);
    int rc = ensure_partials(h, (size_t)grid * (NS + 1));
    if (rc) return rc;
    HypArgs<NO> ha;
    memset(&ha, 0, sizeof(ha));
    ha.n_o = n_o;
    make_exp_args(model, exp, outcomes[0], &ha.base);
    for (int o = 0; o < n_o; ++o) {
        ExpArgs tmp;
        make_exp_args(model, exp, outcomes[o], &tmp);
        ha.comb[o] = tmp.comb;
        ha.log_comb[o] = tmp.log_comb;
        ha.outcome[o] = outcomes[o];
    }
    if (shift) for (int m = 0; m < model->d && m < QSMC_MAX_D; ++m) ha.shift[m] = shift[m];
    const ReduceOut ro = make_reduce(h, true, nullptr);
    hipLaunchKernelGGL((k_hyp_sums<KIND, NO>), dim3(grid), dim3(QSMC_BLOCK), 0, s, x, ldx, n, w, norm, ha, ro);
    HIP_TRY(h, hipGetLastError());
    rc = launch_reduce(h, NS, grid, ro, s);
    if (rc) return rc;
    rc = wait_reduction(h, s);
    if (rc) return rc;
    memcpy(out_host, h->mapped, (size_t)n_o * PER * sizeof(double));
    return QSMC_OK;
}

template <int KIND>
static int hyp_dispatch(qsmc_ctx *h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                        const double *w, double norm, const qsmc_expparam_t *exp, const int64_t *outcomes,
                        int n_o, const double *shift, double *out_host, hipStream_t s) {
    // outcome lists longer than 32 (binomial with n_meas > 31) are processed in groups
    constexpr int D = Model<KIND>::D <= 4 ? Model<KIND>::D : 0;
    constexpr int PER = 2 + 2 * D;
    constexpr bool WIDE = PER * 32 <= 128;      // 32-outcome instantiation only where it fits in registers
    int done = 0;
    while (done < n_o) {
        const int m = n_o - done;
        int take, rc;
        if (m <= 2) {
            take = m;
            rc = hyp_launch<KIND, 2>(h, model, x, ldx, n, w, norm, exp, outcomes + done, take, shift,
                                     out_host + (size_t)done * PER, s);
        } else if (WIDE && m > 8) {
            take = m < 32 ? m : 32;
            if constexpr (WIDE)
                rc = hyp_launch<KIND, 32>(h, model, x, ldx, n, w, norm, exp, outcomes + done, take, shift,
                                          out_host + (size_t)done * PER, s);
            else
                rc = QSMC_ERR_INVALID;
        } else {
            take = m < 8 ? m : 8;
            rc = hyp_launch<KIND, 8>(h, model, x, ldx, n, w, norm, exp, outcomes + done, take, shift,
                                     out_host + (size_t)done * PER, s);
        }
        if (rc) return rc;
        done += take;
    }
    return QSMC_OK;
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int qsmc_abi_version(void) { return QSMC_ABI_VERSION; }

const char *qsmc_strerror(int status) {
    switch (status) {
        case QSMC_OK: return "ok";
        case QSMC_ERR_INVALID: return "invalid argument";
        case QSMC_ERR_HIP: return "HIP runtime error";
        case QSMC_ERR_ALLOC: return "allocation failed";
        case QSMC_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown status";
    }
}

const char *qsmc_last_hip_error(qsmc_handle_t h) { return h ? h->hip_err : ""; }

int qsmc_create(qsmc_handle_t *out, int device) {
    if (!out) return QSMC_ERR_INVALID;
    qsmc_ctx *h = new (std::nothrow) qsmc_ctx();
    if (!h) return QSMC_ERR_ALLOC;
    memset(h, 0, sizeof(*h));
    h->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&h->counter, 2 * sizeof(long long));   // [0] failed, [1] retry count
    if (e == hipSuccess) e = hipMemset(h->counter, 0, 2 * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc(&h->gbar, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(h->gbar, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&h->red_out, REDUCE_OUT_MAX * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc(&h->mapped, REDUCE_OUT_MAX * sizeof(double), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&h->mapped_dev, h->mapped, 0);
    if (e == hipSuccess) e = hipHostMalloc(&h->flag, 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&h->flag_dev, h->flag, 0);
    if (e == hipSuccess) *h->flag = 0ull;
    if (e != hipSuccess) {
        delete h;
        return QSMC_ERR_HIP;
    }
    *out = h;
    return QSMC_OK;
}

int qsmc_destroy(qsmc_handle_t h) {
    if (!h) return QSMC_OK;
    if (h->partials) (void)hipFree(h->partials);
    if (h->rs_offsets) (void)hipFree(h->rs_offsets);
    if (h->tile_sums) (void)hipFree(h->tile_sums);
    if (h->sort_tmp) (void)hipFree(h->sort_tmp);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->counter) (void)hipFree(h->counter);
    if (h->gbar) (void)hipFree(h->gbar);
    if (h->iscratch) (void)hipFree(h->iscratch);
    if (h->cdf_scratch) (void)hipFree(h->cdf_scratch);
    if (h->red_out) (void)hipFree(h->red_out);
    if (h->mapped) (void)hipHostFree(h->mapped);
    if (h->flag) (void)hipHostFree(h->flag);
    if (h->prof_ev) {
        for (int i = 0; i < 2 * QSMC_PROF_CAP; ++i) (void)hipEventDestroy(h->prof_ev[i]);
        free(h->prof_ev);
        free(h->prof_tag);
    }
    delete h;
    return QSMC_OK;
}

int qsmc_set_profiling(qsmc_handle_t h, int enabled) {
    if (!h) return QSMC_ERR_INVALID;
    if (enabled && !h->prof_ev) {
        hipEvent_t *ev = static_cast<hipEvent_t *>(calloc(2 * QSMC_PROF_CAP, sizeof(hipEvent_t)));
        if (!ev) return QSMC_ERR_ALLOC;
        for (int i = 0; i < 2 * QSMC_PROF_CAP; ++i) HIP_TRY(h, hipEventCreate(&ev[i]));
        h->prof_ev = ev;
        h->prof_tag = static_cast<unsigned char *>(calloc(QSMC_PROF_CAP, 1));
        if (!h->prof_tag) return QSMC_ERR_ALLOC;
    }
    h->profiling = enabled ? 1 : 0;
    h->prof_stride = enabled > 1 ? enabled : 1;
    h->prof_n = 0;
    memset(h->prof_seen, 0, sizeof(h->prof_seen));
    return QSMC_OK;
}

int qsmc_last_update_kernel_ms(qsmc_handle_t h, float *ms_out) {
    if (!h || !ms_out || !h->prof_ev || h->prof_n < 1) return QSMC_ERR_INVALID;
    int slot = -1;
    for (int i = h->prof_n - 1; i >= 0 && i > h->prof_n - 1 - QSMC_PROF_CAP; --i)
        if (h->prof_tag[i % QSMC_PROF_CAP] != QSMC_PROF_SAMPLE) { slot = i % QSMC_PROF_CAP; break; }
    if (slot < 0) return QSMC_ERR_INVALID;
    HIP_TRY(h, hipEventSynchronize(h->prof_ev[2 * slot + 1]));
    HIP_TRY(h, hipEventElapsedTime(ms_out, h->prof_ev[2 * slot], h->prof_ev[2 * slot + 1]));
    return QSMC_OK;
}

int qsmc_profile_read(qsmc_handle_t h, float *ms_out, int32_t *tags_out, int32_t cap, int32_t *n_out) {
    if (!h || !ms_out || !n_out || cap < 0) return QSMC_ERR_INVALID;
    int n = h->prof_n < QSMC_PROF_CAP ? h->prof_n : QSMC_PROF_CAP;
    if (n > cap) n = cap;
    const int first = h->prof_n - n;                     // oldest launch still in the ring (or wanted)
    for (int i = 0; i < n; ++i) {
        const int slot = (first + i) % QSMC_PROF_CAP;
        HIP_TRY(h, hipEventSynchronize(h->prof_ev[2 * slot + 1]));
        HIP_TRY(h, hipEventElapsedTime(&ms_out[i], h->prof_ev[2 * slot], h->prof_ev[2 * slot + 1]));
        if (tags_out) tags_out[i] = h->prof_tag[slot];
    }
    *n_out = n;
    h->prof_n = 0;
    return QSMC_OK;
}

int qsmc_likelihood(qsmc_handle_t h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                    const qsmc_expparam_t *exps, int32_t n_e, const int64_t *outcomes, int32_t n_o,
                    double *L_out, qsmc_stream_t stream) {
    if (!h || !x || !exps || !outcomes || !L_out || n < 0 || n_e < 0 || n_o < 0) return QSMC_ERR_INVALID;
    int rc = check_model(model);
    if (rc) return rc;
    if (n == 0) return QSMC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n, QSMC_BLOCK);
    for (int o = 0; o < n_o; ++o)
        for (int e = 0; e < n_e; ++e) {
            ExpArgs ea;
            make_exp_args(model, &exps[e], outcomes[o], &ea);
            double *L = L_out + ((size_t)o * n_e + e) * (size_t)n;
            switch (model->kind) {
#define LAUNCH_L(K)                                                                                   \
    case K:                                                                                           \
        hipLaunchKernelGGL((k_likelihood<K>), dim3(grid), dim3(QSMC_BLOCK), 0, s, x, ldx, n, ea,       \
                           outcomes[o], L);                                                           \
        break;
                LAUNCH_L(QSMC_MODEL_PRECESSION)
                LAUNCH_L(QSMC_MODEL_BINOMIAL_PRECESSION)
                LAUNCH_L(QSMC_MODEL_RB)
                LAUNCH_L(QSMC_MODEL_RB_INTERLEAVED)
                LAUNCH_L(QSMC_MODEL_BINOMIAL_RB)
                LAUNCH_L(QSMC_MODEL_BINOMIAL_RB_INTERLEAVED)
                LAUNCH_L(QSMC_MODEL_UNKNOWN_T2)
                LAUNCH_L(QSMC_MODEL_TOMOGRAPHY)
#undef LAUNCH_L
            }
        }
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

int qsmc_are_models_valid(qsmc_handle_t h, const qsmc_model_t *model, const double *x, int64_t ldx,
                          int64_t n, uint8_t *valid_out, qsmc_stream_t stream) {
    if (!h || !x || !valid_out || n < 0) return QSMC_ERR_INVALID;
    int rc = check_model(model);
    if (rc) return rc;
    if (n == 0) return QSMC_OK;
    hipLaunchKernelGGL(k_valid, dim3(grid_for(n, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, (hipStream_t)stream, x,
                       ldx, n, model->kind, model->d, model->min_freq, valid_out);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

int qsmc_update_fused(qsmc_handle_t h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                      const double *w_in, double *w_out, double prev_norm, const qsmc_expparam_t *exp,
                      int64_t outcome, double *stats_dev, qsmc_update_stats_t *stats_host, double *moments_host,
                      qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !x || !w_out || !exp || n <= 0) return QSMC_ERR_INVALID;      // w_in == NULL: all-ones weights
    int rc = check_model(model);
    if (rc) return rc;
    const int d = model->d;
    const int dmom = d <= 4 ? d : 0;
    if (moments_host && !dmom) return QSMC_ERR_UNSUPPORTED;      // d > 4: use qsmc_moments
    const int n_mom = dmom + dmom * (dmom + 1) / 2;
    const int ns = 3 + n_mom;
    hipStream_t s = (hipStream_t)stream;
    const bool vec2 = aligned16(x) && (!w_in || aligned16(w_in)) && aligned16(w_out) && (ldx % 2 == 0);
    const int per_block = QSMC_BLOCK * (vec2 ? 2 : 1) * UPD_UNROLL;
    const int grid = grid_for(n, per_block);
    rc = ensure_partials(h, (size_t)grid * (ns + 1));
    if (rc) return rc;
    ExpArgs ea;
    make_exp_args(model, exp, outcome, &ea);
    ReduceOut ro = make_reduce(h, stats_host || moments_host, stats_dev);
    // per-tile sums of the new weights: a resample that follows this update takes its chunk sums from them
    static const bool tile_sums_on = getenv("QSMC_NO_TILE_SUMS") == nullptr;     // (A/B switch for measurements)
    if (tile_sums_on && BUCKET_CHUNK % per_block == 0 &&
        ensure_tile_sums(h, (size_t)((n + per_block - 1) / per_block) * QSMC_WAVES_PER_BLOCK) == QSMC_OK) {
        ro.tile_sums = h->tile_sums;
        h->ts.w = w_out;
        h->ts.n = n;
        h->ts.tile = per_block;
    } else {
        h->ts.w = nullptr;
    }
    ++h->ts.gen;
    h->ts.armed = 0;
    switch (model->kind) {
#define LAUNCH_U(K)                                                                             \
    case K:                                                                                     \
        launch_update<K>(h, vec2, grid, s, x, ldx, n, w_in, w_out, prev_norm, ea, outcome, ro); \
        break;
        LAUNCH_U(QSMC_MODEL_PRECESSION)
        LAUNCH_U(QSMC_MODEL_BINOMIAL_PRECESSION)
        LAUNCH_U(QSMC_MODEL_RB)
        LAUNCH_U(QSMC_MODEL_RB_INTERLEAVED)
        LAUNCH_U(QSMC_MODEL_BINOMIAL_RB)
        LAUNCH_U(QSMC_MODEL_BINOMIAL_RB_INTERLEAVED)
        LAUNCH_U(QSMC_MODEL_UNKNOWN_T2)
        LAUNCH_U(QSMC_MODEL_TOMOGRAPHY)
#undef LAUNCH_U
    }
    HIP_TRY(h, hipGetLastError());
    rc = launch_reduce(h, ns, grid, ro, s);
    if (rc) return rc;
    return collect_stats(h, ns, stats_host, moments_host, n_mom, s);
}

int qsmc_update_multi(qsmc_handle_t h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                      const double *w_in, double *w_out, double prev_norm, const qsmc_expparam_t *exps,
                      const int64_t *outcomes, int32_t k, qsmc_update_stats_t *stats_host, double *moments_host,
                      qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !x || !w_out || !exps || !outcomes || !stats_host || n <= 0 || k < 1 || k > MULTI_KMAX)
        return QSMC_ERR_INVALID;
    int rc = check_model(model);
    if (rc) return rc;
    const int d = model->d;
    const int dmom = d <= 4 ? d : 0;
    if (moments_host && !dmom) return QSMC_ERR_UNSUPPORTED;
    const int n_mom = dmom + dmom * (dmom + 1) / 2;
    const int ns = 3 * MULTI_KMAX + n_mom;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n, QSMC_BLOCK * 4);
    rc = ensure_partials(h, (size_t)grid * (ns + 1));
    if (rc) return rc;
    MultiArgs ma;
    memset(&ma, 0, sizeof(ma));
    ma.k = k;
    for (int j = 0; j < k; ++j) {
        make_exp_args(model, &exps[j], outcomes[j], &ma.e[j]);
        ma.outcome[j] = outcomes[j];
    }
    const ReduceOut ro = make_reduce(h, true, nullptr);
    switch (model->kind) {
#define LAUNCH_MU(K)                                                                                   \
    case K:                                                                                            \
        if (ma.e[0].lik_pow != 0.0)                                                                    \
            hipLaunchKernelGGL((k_update_multi<K, true>), dim3(grid), dim3(QSMC_BLOCK), 0, s, x, ldx, n, w_in, \
                               w_out, prev_norm, ma, ro);                                              \
        else                                                                                           \
            hipLaunchKernelGGL((k_update_multi<K, false>), dim3(grid), dim3(QSMC_BLOCK), 0, s, x, ldx, n, w_in, \
                               w_out, prev_norm, ma, ro);                                              \
        break;
        LAUNCH_MU(QSMC_MODEL_PRECESSION)
        LAUNCH_MU(QSMC_MODEL_BINOMIAL_PRECESSION)
        LAUNCH_MU(QSMC_MODEL_RB)
        LAUNCH_MU(QSMC_MODEL_RB_INTERLEAVED)
        LAUNCH_MU(QSMC_MODEL_BINOMIAL_RB)
        LAUNCH_MU(QSMC_MODEL_BINOMIAL_RB_INTERLEAVED)
        LAUNCH_MU(QSMC_MODEL_UNKNOWN_T2)
        LAUNCH_MU(QSMC_MODEL_TOMOGRAPHY)
#undef LAUNCH_MU
    }
    HIP_TRY(h, hipGetLastError());
    rc = launch_reduce(h, ns, grid, ro, s);
    if (rc) return rc;
    rc = wait_reduction(h, s);
    if (rc) return rc;
    for (int j = 0; j < k; ++j) {
        stats_host[j].sum = h->mapped[3 * j];
        stats_host[j].sumsq = h->mapped[3 * j + 1];
        stats_host[j].n_bad = h->mapped[3 * j + 2];
        stats_host[j].min = h->mapped[ns];
    }
    if (moments_host) memcpy(moments_host, h->mapped + 3 * MULTI_KMAX, (size_t)n_mom * sizeof(double));
    return QSMC_OK;
}

int qsmc_hypothetical_sums(qsmc_handle_t h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                           const double *w, double norm, const qsmc_expparam_t *exp, const int64_t *outcomes,
                           int32_t n_o, const double *shift, double *out_host, qsmc_stream_t stream) {
    if (!h || !x || !exp || !outcomes || !out_host || n <= 0 || n_o < 1) return QSMC_ERR_INVALID;
    int rc = check_model(model);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (model->kind) {
#define HD(K) case K: return hyp_dispatch<K>(h, model, x, ldx, n, w, norm, exp, outcomes, n_o, shift, out_host, s);
        HD(QSMC_MODEL_PRECESSION)
        HD(QSMC_MODEL_BINOMIAL_PRECESSION)
        HD(QSMC_MODEL_RB)
        HD(QSMC_MODEL_RB_INTERLEAVED)
        HD(QSMC_MODEL_BINOMIAL_RB)
        HD(QSMC_MODEL_BINOMIAL_RB_INTERLEAVED)
        HD(QSMC_MODEL_UNKNOWN_T2)
        HD(QSMC_MODEL_TOMOGRAPHY)
#undef HD
    }
    return QSMC_ERR_INVALID;
}

int qsmc_update_from_likelihood(qsmc_handle_t h, const double *L, int64_t n, const double *w_in,
                                double *w_out, double prev_norm, double *stats_dev,
                                qsmc_update_stats_t *stats_host, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !L || !w_in || !w_out || n <= 0) return QSMC_ERR_INVALID;
    return weights_pass<0>(h, L, n, w_in, w_out, prev_norm, stats_dev, stats_host, (hipStream_t)stream);
}

int qsmc_clip_weights(qsmc_handle_t h, double *w, int64_t n, double norm, double *stats_dev,
                      qsmc_update_stats_t *stats_host, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !w || n <= 0) return QSMC_ERR_INVALID;
    return weights_pass<1>(h, nullptr, n, w, w, norm, stats_dev, stats_host, (hipStream_t)stream);
}

int qsmc_weight_stats(qsmc_handle_t h, const double *w, int64_t n, double norm, double *stats_dev,
                      qsmc_update_stats_t *stats_host, qsmc_stream_t stream) {
    if (!h || !w || n <= 0) return QSMC_ERR_INVALID;
    return weights_pass<3>(h, nullptr, n, w, nullptr, norm, stats_dev, stats_host, (hipStream_t)stream);
}

int qsmc_weight_entropy(qsmc_handle_t h, const double *w, int64_t n, double norm, double *entropy_host,
                        qsmc_stream_t stream) {
    if (!h || n <= 0 || !entropy_host || !(norm > 0.0)) return QSMC_ERR_INVALID;
    if (!w) {                                    // implicit uniform weights 1 / n_total with norm = n_total
        *entropy_host = (double)n / norm * log(norm);
        return QSMC_OK;
    }
    qsmc_update_stats_t st;
    const int rc = weights_pass<4>(h, nullptr, n, w, nullptr, norm, nullptr, &st, (hipStream_t)stream);
    if (rc) return rc;
    *entropy_host = st.sumsq;
    return QSMC_OK;
}

int qsmc_normalize_weights(qsmc_handle_t h, const double *w_in, double *w_out, int64_t n, double norm,
                           qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !w_in || !w_out || n < 0) return QSMC_ERR_INVALID;
    if (n == 0) return QSMC_OK;
    return weights_pass<2>(h, nullptr, n, w_in, w_out, norm, nullptr, nullptr, (hipStream_t)stream);
}

int qsmc_fill(qsmc_handle_t h, double *w, int64_t n, double value, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !w || n < 0) return QSMC_ERR_INVALID;
    if (n == 0) return QSMC_OK;
    hipLaunchKernelGGL(k_fill, dim3(grid_for(n, QSMC_BLOCK * 4)), dim3(QSMC_BLOCK), 0, (hipStream_t)stream, w,
                       n, value);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

int qsmc_moments(qsmc_handle_t h, const double *x, int64_t ldx, int64_t n, int32_t d, const double *w,
                 double norm, double *out_dev, double *out_host, qsmc_stream_t stream) {
    if (!h || !x || !w || n <= 0 || d < 1 || d > QSMC_MAX_D) return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int K = 1 + d + d * (d + 1) / 2;
    int rc = ensure_scratch(h, 256 + (size_t)QSMC_MAX_D * (2 + QSMC_MAX_D));
    if (rc) return rc;
    double *dst = out_dev ? out_dev : h->scratch;
    const int grid = grid_for(n, QSMC_BLOCK * 4);
    if (d <= 4) {
        rc = ensure_partials(h, (size_t)grid * (K + 1));
        if (rc) return rc;
        const ReduceOut ro = make_reduce(h, out_host != nullptr, nullptr);
        switch (d) {
#define LAUNCH_M(DD)                                                                                  \
    case DD:                                                                                          \
        hipLaunchKernelGGL((k_moments_small<DD>), dim3(grid), dim3(QSMC_BLOCK), 0, s, x, ldx, n, w, norm, ro); \
        break;
            LAUNCH_M(1) LAUNCH_M(2) LAUNCH_M(3) LAUNCH_M(4)
#undef LAUNCH_M
        }
        HIP_TRY(h, hipGetLastError());
        rc = launch_reduce(h, K, grid, ro, s);
        if (rc) return rc;
        if (out_dev)
            HIP_TRY(h, hipMemcpyAsync(out_dev, h->red_out, K * sizeof(double), hipMemcpyDeviceToDevice, s));
        if (out_host) {
            rc = wait_reduction(h, s);
            if (rc) return rc;
            memcpy(out_host, h->mapped, K * sizeof(double));
        }
        return QSMC_OK;
    }
    // d > 4: X diag(w) X^T on the f64 matrix cores (k_moments_mfma), then a one-workgroup sum of partials
    const int gridm = grid_for(n, QSMC_BLOCK) < 1024 ? grid_for(n, QSMC_BLOCK) : 1024;   // 64 particles / wave tile
    rc = ensure_partials(h, (size_t)gridm * MFMA_MOM_K);
    if (rc) return rc;
    rc = ensure_scratch(h, 256 + MFMA_MOM_K);
    if (rc) return rc;
    hipLaunchKernelGGL(k_moments_mfma, dim3(gridm), dim3(QSMC_BLOCK), 0, s, x, ldx, n, d, w, norm, h->partials);
    double *full = h->scratch + 256;
    hipLaunchKernelGGL(k_sum_partials, dim3((MFMA_MOM_K + QSMC_WAVES_PER_BLOCK - 1) / QSMC_WAVES_PER_BLOCK), dim3(QSMC_BLOCK), 0, s,
                       h->partials, gridm, MFMA_MOM_K, full);
    HIP_TRY(h, hipGetLastError());
    double hostfull[MFMA_MOM_K];
    rc = read_back(h, full, hostfull, MFMA_MOM_K, s);
    if (rc) return rc;
    double packed[1 + QSMC_MAX_D + QSMC_MAX_D * (QSMC_MAX_D + 1) / 2];
    packed[0] = hostfull[272];
    int k = 1 + d;
    for (int m = 0; m < d; ++m) {
        packed[1 + m] = hostfull[256 + m];
        for (int q = m; q < d; ++q) packed[k++] = hostfull[m * 16 + q];
    }
    if (out_host) memcpy(out_host, packed, K * sizeof(double));
    if (out_dev) {
        HIP_TRY(h, hipMemcpyAsync(out_dev, packed, K * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    return QSMC_OK;
}

int qsmc_cumsum(qsmc_handle_t h, const double *w, int64_t n, double norm, double *cdf,
                qsmc_stream_t stream) {
    if (!h || !w || !cdf || n <= 0) return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int64_t chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    int rc = ensure_partials(h, (size_t)chunks + 1);
    if (rc) return rc;
    const double inv_norm = 1.0 / norm;
    hipLaunchKernelGGL(k_chunk_sums, dim3((unsigned)chunks), dim3(QSMC_BLOCK), 0, s, w, n, inv_norm, h->partials);
    if (chunks > (int64_t)SCAN_SUMS_THREADS * SCAN_SUMS_MAX_PER)
        hipLaunchKernelGGL(k_scan_sums_big, dim3(1), dim3(QSMC_BLOCK), 0, s, h->partials, chunks,
                           (unsigned long long *)nullptr, TileSrc{nullptr, 0, 0, 0.0});
    else
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_SUMS_THREADS), 0, s, h->partials, chunks,
                           (unsigned long long *)nullptr, TileSrc{nullptr, 0, 0, 0.0});
    hipLaunchKernelGGL(k_chunk_scan, dim3((unsigned)chunks), dim3(SCAN_THREADS), 0, s, w, n, inv_norm, h->partials,
                       cdf, (const unsigned long long *)nullptr);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

int qsmc_lw_ancestors(qsmc_handle_t h, const double *cdf, int64_t n_in, const double *u, int64_t n_out,
                      int64_t *js, qsmc_stream_t stream) {
    if (!h || !cdf || !u || !js || n_in <= 0 || n_out < 0) return QSMC_ERR_INVALID;
    if (n_out == 0) return QSMC_OK;
    hipLaunchKernelGGL(k_ancestors, dim3(grid_for(n_out, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0,
                       (hipStream_t)stream, cdf, n_in, u, n_out, js);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

static void fill_lw(LWArgs *lw, int d, double a, const double *mean, const double *S) {
    memset(lw, 0, sizeof(*lw));
    lw->a = a;
    if (mean) for (int m = 0; m < d; ++m) lw->mean[m] = mean[m];
    if (S) for (int k = 0; k < d * d; ++k) lw->S[k] = S[k];
}

int qsmc_lw_centres(qsmc_handle_t h, const double *x_in, int64_t ldx_in, int32_t d, const int64_t *js,
                    int64_t n_out, double a, const double *mean, double *mus, int64_t ld_mus,
                    qsmc_stream_t stream) {
    if (!h || !x_in || !js || !mean || !mus || d < 1 || d > QSMC_MAX_D || n_out < 0) return QSMC_ERR_INVALID;
    if (n_out == 0) return QSMC_OK;
    LWArgs lw;
    fill_lw(&lw, d, a, mean, nullptr);
    hipLaunchKernelGGL(k_centres, dim3(grid_for(n_out, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, (hipStream_t)stream,
                       x_in, ldx_in, d, js, n_out, a, lw, mus, ld_mus);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

int qsmc_lw_perturb(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect, const double *mus,
                    int64_t ld_mus, const int64_t *idxs, int64_t k, int32_t centre_by_idx, const double *S,
                    const double *z, int64_t ldz, double *x_out, int64_t ldx_out, uint8_t *valid_out,
                    qsmc_stream_t stream) {
    if (!h || !model || !mus || !S || !z || !x_out || !valid_out || k < 0) return QSMC_ERR_INVALID;
    if (model->d < 1 || model->d > QSMC_MAX_D) return QSMC_ERR_INVALID;
    if (k == 0) return QSMC_OK;
    LWArgs lw;
    fill_lw(&lw, model->d, 0.0, nullptr, S);
    hipLaunchKernelGGL(k_perturb, dim3(grid_for(k, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, (hipStream_t)stream,
                       model->kind, model->d, model->min_freq, postselect, mus, ld_mus, idxs, k, centre_by_idx,
                       lw, z, ldz, x_out, ldx_out, valid_out);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

static int read_counter(qsmc_ctx *h, int64_t *out, hipStream_t s) {
    int rc = ensure_pinned(h, 1);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->pinned, h->counter, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    long long v;
    memcpy(&v, h->pinned, sizeof(v));
    *out = (int64_t)v;
    return QSMC_OK;
}

// Layout of the bucketed resampler's integer scratch:
//   hist[256][chunks] | counts[chunks] | slot_off[chunks+1] i64 | item_off[chunks+1] | item_chunk[max_items] |
//   retry_list[n_out] u32
struct BucketPlan {
    bool bucketed;
    int chunks, max_items;
    unsigned int *hist, *counts, *retry_list;
    long long *slot_off;
    int *item_off, *item_chunk;
};

static bool use_buckets(int64_t chunks64, int64_t n_out) {
    static const bool forced_direct = getenv("QSMC_DIRECT_RESAMPLE") != nullptr;     // (test / measurement switch)
    return chunks64 <= BUCKET_MAX_CHUNKS && n_out >= 4 * BUCKET_CHUNK && n_out < (1ll << 32) && !forced_direct;
}

static int bucket_plan_layout(qsmc_ctx *h, int64_t chunks64, int64_t n_out, BucketPlan *bp) {
    bp->bucketed = use_buckets(chunks64, n_out);
    if (!bp->bucketed) return QSMC_OK;
    const int chunks = (int)chunks64;
    const size_t hist_b = (size_t)BUCKET_COUNT_BLOCKS * chunks * sizeof(unsigned int);
    const size_t counts_b = ((size_t)chunks * sizeof(unsigned int) + 15) & ~(size_t)15;
    const size_t slot_b = ((size_t)(chunks + 1) * sizeof(long long) + 15) & ~(size_t)15;
    const size_t item_b = ((size_t)(chunks + 1) * sizeof(int) + 15) & ~(size_t)15;
    const int max_items = chunks + (int)(n_out / BUCKET_CAP) + 1;
    const size_t map_b = ((size_t)max_items * sizeof(int) + 15) & ~(size_t)15;
    const size_t retry_b = (size_t)n_out * sizeof(unsigned int);
    const int rc = ensure_iscratch(h, hist_b + counts_b + slot_b + item_b + map_b + retry_b);
    if (rc) return rc;
    unsigned char *basep = reinterpret_cast<unsigned char *>(h->iscratch);
    bp->chunks = chunks;
    bp->max_items = max_items;
    bp->hist = reinterpret_cast<unsigned int *>(basep);
    bp->counts = reinterpret_cast<unsigned int *>(basep + hist_b);
    bp->slot_off = reinterpret_cast<long long *>(basep + hist_b + counts_b);
    bp->item_off = reinterpret_cast<int *>(basep + hist_b + counts_b + slot_b);
    bp->item_chunk = reinterpret_cast<int *>(basep + hist_b + counts_b + slot_b + item_b);
    bp->retry_list = reinterpret_cast<unsigned int *>(basep + hist_b + counts_b + slot_b + item_b + map_b);
    return QSMC_OK;
}

static void philox_keys(uint64_t seed, uint64_t epoch, uint32_t *k0, uint32_t *k1, uint32_t *ep) {
    *k0 = (uint32_t)seed;
    *k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(epoch >> 16);
    *ep = (uint32_t)(epoch & 0xFFFFu);
}

// The part of a device-RNG resample that needs only the weights: chunk sums -> monotone offsets, the
// zeroed counters and (bucketed sampler) the multinomial chunk counts and the work-item plan.  It can
// be queued the moment the n_ess test fails, before the host has formed mean / covariance / sqrtm.
static int resample_prefix(qsmc_ctx *h, const double *w, int64_t n_in, double norm, int64_t n_out, uint64_t seed,
                           uint64_t epoch, hipStream_t s) {
    uint32_t k0, k1, ep;
    philox_keys(seed, epoch, &k0, &k1, &ep);
    const int64_t chunks64 = (n_in + BUCKET_CHUNK - 1) / BUCKET_CHUNK;
    int rc = ensure_rs_offsets(h, (size_t)chunks64 + 1);
    if (rc) return rc;
    double *offsets = h->rs_offsets;
    const double inv_norm = 1.0 / norm;
    // chunk sums: from the tile sums of the update that produced these very weights, if the caller vouches for
    // that (qsmc_lw_use_update_sums) and nothing has touched them since; else one pass over the weights
    TileSrc ts{nullptr, 0, 0, 0.0};
    if (h->ts.armed && h->ts.armed == h->ts.gen && w && h->ts.w == w && h->ts.n == n_in)
        ts = TileSrc{h->tile_sums, BUCKET_CHUNK / h->ts.tile * QSMC_WAVES_PER_BLOCK,
                     (n_in + h->ts.tile - 1) / h->ts.tile * QSMC_WAVES_PER_BLOCK, inv_norm};
    h->ts.armed = 0;
    BucketPlan bp;
    rc = bucket_plan_layout(h, chunks64, n_out, &bp);
    if (rc) return rc;
    static const bool count_by_draws = getenv("QSMC_COUNT_BY_DRAWS") != nullptr;   // (measurement switch: the
                                                                //  one-uniform-per-output histogram, same law)
    // with tile sums the bucketed count kernel forms the offsets itself; otherwise: chunk sums, then the scan
    const bool scan_in_counts = ts.tiles && bp.bucketed && !count_by_draws;
    if (!ts.tiles)
        hipLaunchKernelGGL(k_chunk_sums, dim3((unsigned)chunks64), dim3(QSMC_BLOCK), 0, s, w, n_in, inv_norm, offsets);
    if (scan_in_counts) {
    } else if (chunks64 > (int64_t)SCAN_SUMS_THREADS * SCAN_SUMS_MAX_PER)
        hipLaunchKernelGGL(k_scan_sums_big, dim3(1), dim3(QSMC_BLOCK), 0, s, offsets, chunks64,
                           reinterpret_cast<unsigned long long *>(h->counter), ts);
    else
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_SUMS_THREADS), 0, s, offsets, chunks64,
                           reinterpret_cast<unsigned long long *>(h->counter), ts);
    if (bp.bucketed) {
        const int chunks = bp.chunks;
        if (count_by_draws) {
            const size_t lds = (size_t)(chunks + (chunks >> 5) + (chunks >> 10) + 8) * sizeof(double) +
                               (size_t)chunks * sizeof(unsigned int) + (size_t)(GUIDE_BINS + 1 + 32) * sizeof(int);
            size_t &lds_granted = h->count_lds_granted;        // the opt-in for > 64 KB of dynamic LDS is sticky: ask once per size
            if (lds_granted < 64 * 1024) lds_granted = 64 * 1024;
            if (lds > lds_granted) {
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_bucket_count),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                lds_granted = lds;
            }
            hipLaunchKernelGGL(k_bucket_count, dim3(BUCKET_COUNT_BLOCKS), dim3(BUCKET_COUNT_THREADS), lds, s, offsets,
                               chunks, n_out, k0, k1, ep, bp.hist);
            hipLaunchKernelGGL(k_bucket_reduce, dim3((chunks + QSMC_WAVE - 1) / QSMC_WAVE), dim3(QSMC_BLOCK), 0, s,
                               bp.hist, BUCKET_COUNT_BLOCKS, chunks, bp.counts);
            hipLaunchKernelGGL(k_bucket_plan, dim3(1), dim3(1024), 0, s, bp.counts, chunks, bp.slot_off, bp.item_off,
                               bp.item_chunk);
        } else {
            const char *margin_env = getenv("QSMC_POISSON_MARGIN");          // (test switch: 0 makes the removal branch common)
            const double kappa = margin_env ? atof(margin_env) : 5.0;
            double lambda = (double)n_out - kappa * sqrt((double)n_out);
            if (!(lambda > 0.0)) lambda = 0.0;
            // LDS: 8192 chunks need 68 KB of edges + 32 KB of counters: the opt-in beyond 64 KB is sticky
            const size_t lds = (size_t)(chunks + (chunks >> 5) + (chunks >> 10) + 8) * sizeof(double) +
                               (size_t)chunks * sizeof(unsigned int);
            size_t &lds_granted = h->topup_lds_granted;
            if (lds_granted < 48 * 1024) lds_granted = 48 * 1024;
            if (lds > lds_granted) {
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_bucket_counts),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                lds_granted = lds;
            }
            unsigned int *extra = bp.hist;                       // (the histogram rows are not used on this path)
            hipLaunchKernelGGL(k_bucket_counts, dim3(BUCKET_COUNTS_BLOCKS), dim3(BUCKET_COUNTS_THREADS), lds, s, offsets,
                               scan_in_counts ? ts : TileSrc{nullptr, 0, 0, 0.0},
                               reinterpret_cast<unsigned long long *>(h->counter), chunks, n_out, lambda, k0, k1, ep,
                               bp.counts, extra, bp.slot_off, bp.item_off, bp.item_chunk, h->gbar, h->gbar_base);
            h->gbar_base += 2ull * BUCKET_COUNTS_BLOCKS;
        }
    }
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

static int resample_philox_impl(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect,
                                const double *x_in, int64_t ldx_in, int64_t n_in, int32_t d, const double *w,
                                double norm, double a, const double *mean, const double *S, int64_t n_out,
                                uint64_t seed, uint64_t epoch, int32_t maxiter, double *x_out, const OutPlace &pl,
                                int64_t *n_failed_host, qsmc_stream_t stream) {
    if (!h || !model || !x_in || !mean || !S || !x_out || n_in <= 0 || n_out <= 0) return QSMC_ERR_INVALID;
    if (d != model->d || d < 1 || d > QSMC_MAX_D || maxiter < 1) return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    LWArgs lw;
    fill_lw(&lw, d, a, mean, S);
    uint32_t k0, k1, ep;
    philox_keys(seed, epoch, &k0, &k1, &ep);
    const int64_t chunks64 = (n_in + BUCKET_CHUNK - 1) / BUCKET_CHUNK;
    const bool prepared = h->prep.valid && h->prep.w == w && h->prep.n_in == n_in && h->prep.n_out == n_out &&
                          h->prep.norm == norm && h->prep.seed == seed && h->prep.epoch == epoch &&
                          h->prep.stream == s;
    h->prep.valid = 0;
    int rc = QSMC_OK;
    if (!prepared) {
        rc = resample_prefix(h, w, n_in, norm, n_out, seed, epoch, s);
        if (rc) return rc;
    }
    double *offsets = h->rs_offsets;
    const double inv_norm = 1.0 / norm;
    unsigned long long *nf = reinterpret_cast<unsigned long long *>(h->counter);
    unsigned long long *retry_count = nf + 1;
    rc = ensure_cdf(h, (size_t)n_in);
    if (rc) return rc;
    BucketPlan bp;
    rc = bucket_plan_layout(h, chunks64, n_out, &bp);      // no reallocation: the prefix sized it
    if (rc) return rc;
    if (!bp.bucketed) {
        // small or very large clouds: materialise the CDF and search it directly
        hipLaunchKernelGGL(k_chunk_scan, dim3((unsigned)chunks64), dim3(SCAN_THREADS), 0, s, w, n_in, inv_norm, offsets,
                           h->cdf_scratch, (const unsigned long long *)nullptr);
        hipLaunchKernelGGL(k_resample_philox, dim3(grid_for(n_out, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, s,
                           model->kind, d, model->min_freq, postselect, x_in, ldx_in, n_in, h->cdf_scratch, lw, n_out,
                           k0, k1, ep, maxiter, x_out, pl, nf);
    } else {
        const int chunks = bp.chunks;
        // 512-thread workgroups, CDF chunk + guide in LDS (40 KB -> 3 resident workgroups per CU, so one
        // workgroup's scan phase overlaps another's sampling loop); x is gathered from the chunk's
        // 32 KB global window (L2-resident).
        hipEvent_t pe0 = nullptr, pe1 = nullptr;
        prof_events(h, QSMC_PROF_SAMPLE, &pe0, &pe1);
#define LAUNCH_B(DD, BT)                                                                                       \
    hipExtLaunchKernelGGL((k_bucket_sample<DD, BT>), dim3(bp.max_items), dim3(BT), 0, s, pe0, pe1, 0, model->kind, d, \
                       model->min_freq, postselect, x_in, ldx_in, n_in, w, inv_norm, offsets,                       \
                       chunks, bp.slot_off, bp.item_off, bp.item_chunk, lw, k0, k1, ep,                             \
                       maxiter, x_out, pl, nf, bp.retry_list, retry_count)
        switch (d) {
            case 1: LAUNCH_B(1, 512); break;
            case 2: LAUNCH_B(2, 512); break;
            case 3: LAUNCH_B(3, 512); break;
            case 4: LAUNCH_B(4, 512); break;
            case 16: LAUNCH_B(16, 512); break;         // 2-qubit tomography: all indices static
            default: LAUNCH_B(0, 512); break;          // other d up to 16: runtime-d kernel
        }
#undef LAUNCH_B
        if (postselect && maxiter > 1) {
            // only if some particle asked for a global redraw do these two do any work
            // (more than two processes on one GPU -- bench.py's control-flow check -- must shrink the grid: all of
            //  them have to be resident together, 512 workgroup slots in total)
            static const int redraw_blocks = [] {
                const char *e = getenv("QSMC_REDRAW_BLOCKS");
                const int v = e ? atoi(e) : REDRAW_BLOCKS;
                return v < 1 ? 1 : (v > REDRAW_BLOCKS ? REDRAW_BLOCKS : v);
            }();
            hipLaunchKernelGGL((d <= 4 ? k_bucket_redraw<4> : k_bucket_redraw<QSMC_MAX_D>), dim3(redraw_blocks),
                               dim3(SCAN_THREADS), 0, s, model->kind, d,
                               model->min_freq, x_in, ldx_in, n_in, w, inv_norm, offsets, chunks64, h->cdf_scratch, lw,
                               k0, k1, ep, maxiter, x_out, pl, bp.retry_list, retry_count, nf, h->gbar + 2);
        }
    }
    HIP_TRY(h, hipGetLastError());
    if (n_failed_host) return read_counter(h, n_failed_host, s);
    // asynchronous form: the count reaches pinned memory with the next host-visible reduction (every
    // update publishes it, see k_reduce_partials) or on demand (qsmc_last_resample_failed, synchronize = 1)
    return QSMC_OK;
}

int qsmc_lw_use_update_sums(qsmc_handle_t h, uint64_t update_token) {
    if (!h) return QSMC_ERR_INVALID;
    h->ts.armed = update_token;
    return QSMC_OK;
}

int qsmc_update_token(qsmc_handle_t h, uint64_t *token_out) {
    if (!h || !token_out) return QSMC_ERR_INVALID;
    *token_out = h->ts.gen;
    return QSMC_OK;
}

int qsmc_lw_resample_prepare(qsmc_handle_t h, const double *w, int64_t n_in, double norm, int64_t n_out,
                             uint64_t seed, uint64_t epoch, qsmc_stream_t stream) {
    if (!h || n_in <= 0 || n_out <= 0 || !(norm > 0.0)) return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    h->prep.valid = 0;
    const int rc = resample_prefix(h, w, n_in, norm, n_out, seed, epoch, s);
    if (rc) return rc;
    h->prep.valid = 1;
    h->prep.w = w;
    h->prep.n_in = n_in;
    h->prep.n_out = n_out;
    h->prep.norm = norm;
    h->prep.seed = seed;
    h->prep.epoch = epoch;
    h->prep.stream = s;
    return QSMC_OK;
}

int qsmc_lw_resample_philox(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect,
                            const double *x_in, int64_t ldx_in, int64_t n_in, int32_t d, const double *w,
                            double norm, double a, const double *mean, const double *S, int64_t n_out,
                            uint64_t seed, uint64_t epoch, int32_t maxiter, double *x_out, int64_t ldx_out,
                            int64_t *n_failed_host, qsmc_stream_t stream) {
    OutPlace pl;
    memset(&pl, 0, sizeof(pl));
    pl.ld_m = ldx_out;
    pl.ld_s = 1;
    return resample_philox_impl(h, model, postselect, x_in, ldx_in, n_in, d, w, norm, a, mean, S, n_out, seed,
                                epoch, maxiter, x_out, pl, n_failed_host, stream);
}

int qsmc_lw_resample_philox_sharded(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect,
                                    const double *x_in, int64_t ldx_in, int64_t n_in, int32_t d,
                                    const double *w, double norm, double a, const double *mean, const double *S,
                                    const int64_t *dest_counts, int32_t n_dest, uint64_t seed, uint64_t epoch,
                                    int32_t maxiter, double *rows_out, int64_t *n_failed_host,
                                    qsmc_stream_t stream) {
    if (!dest_counts || n_dest < 1 || n_dest > QSMC_MAX_DEST) return QSMC_ERR_INVALID;
    OutPlace pl;
    memset(&pl, 0, sizeof(pl));
    pl.n_dest = n_dest;
    pl.ld_m = 1;                 // AoS rows [n_out][d], grouped by destination rank
    pl.ld_s = d;
    int64_t n_out = 0;
    for (int r = 0; r < n_dest; ++r) {
        if (dest_counts[r] < 0) return QSMC_ERR_INVALID;
        pl.dest_base[r] = n_out;
        n_out += dest_counts[r];
        pl.order[r] = r;
    }
    for (int i = 1; i < n_dest; ++i)                       // insertion sort by quota (stable)
        for (int j = i; j > 0 && dest_counts[pl.order[j - 1]] > dest_counts[pl.order[j]]; --j) {
            const int t = pl.order[j];
            pl.order[j] = pl.order[j - 1];
            pl.order[j - 1] = t;
        }
    int64_t start = 0, prev = 0;
    for (int sidx = 0; sidx < n_dest; ++sidx) {
        pl.quota[sidx] = dest_counts[pl.order[sidx]];
        pl.seg_start[sidx] = start;
        start += (pl.quota[sidx] - prev) * (int64_t)(n_dest - sidx);
        prev = pl.quota[sidx];
    }
    pl.seg_start[n_dest] = start;
    if (n_out == 0) return QSMC_OK;
    return resample_philox_impl(h, model, postselect, x_in, ldx_in, n_in, d, w, norm, a, mean, S, n_out, seed,
                                epoch, maxiter, rows_out, pl, n_failed_host, stream);
}

int qsmc_last_resample_failed(qsmc_handle_t h, int64_t *n_failed_out, int32_t synchronize, qsmc_stream_t stream) {
    if (!h || !n_failed_out) return QSMC_ERR_INVALID;
    if (synchronize) {
        hipLaunchKernelGGL(k_publish_counter, dim3(1), dim3(64), 0, (hipStream_t)stream,
                           reinterpret_cast<const unsigned long long *>(h->counter),
                           h->mapped_dev + (REDUCE_OUT_MAX - 1));
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    }
    *n_failed_out = (int64_t)h->mapped[REDUCE_OUT_MAX - 1];
    return QSMC_OK;
}

int qsmc_prior_uniform_philox(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect,
                              const double *lo, const double *hi, int32_t d, int64_t n, uint64_t seed,
                              uint64_t epoch, int32_t maxiter, double *x_out, int64_t ldx_out,
                              int64_t *n_failed_host, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !model || !lo || !hi || !x_out || n <= 0 || d < 1 || d > QSMC_MAX_D || maxiter < 1)
        return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    LWArgs box;
    memset(&box, 0, sizeof(box));
    for (int m = 0; m < d; ++m) {
        box.mean[m] = lo[m];
        box.S[m] = hi[m] - lo[m];
    }
    HIP_TRY(h, hipMemsetAsync(h->counter, 0, sizeof(long long), s));
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(epoch >> 16);
    hipLaunchKernelGGL(k_prior_uniform_philox, dim3(grid_for(n, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, s,
                       model->kind, d, model->min_freq, postselect, box, n, k0, k1,
                       (uint32_t)(epoch & 0xFFFFu), maxiter, x_out, ldx_out,
                       reinterpret_cast<unsigned long long *>(h->counter));
    HIP_TRY(h, hipGetLastError());
    if (n_failed_host) return read_counter(h, n_failed_host, s);
    return QSMC_OK;
}

int qsmc_random_walk(qsmc_handle_t h, double *x, int64_t ldx, int64_t n, int32_t d, const double *scale,
                     const double *z, int64_t ldz, uint64_t seed, uint64_t epoch, qsmc_stream_t stream) {
    if (!h || !x || !scale || n < 0 || d < 1 || d > QSMC_MAX_D) return QSMC_ERR_INVALID;
    WalkArgs wa;
    memset(&wa, 0, sizeof(wa));
    for (int m = 0; m < d; ++m) {
        if (!(scale[m] == scale[m])) return QSMC_ERR_INVALID;
        if (scale[m] != 0.0) {
            wa.scale[wa.n_rw] = scale[m];
            wa.row[wa.n_rw] = m;
            ++wa.n_rw;
        }
    }
    if (n == 0 || wa.n_rw == 0) return QSMC_OK;
    if (z && wa.n_rw > 1 && ldz < n) return QSMC_ERR_INVALID;      // (a single row's stride is never used)
    // (moves particles, not weights: a queued resample prefix stays valid)
    const uint32_t k0 = (uint32_t)seed ^ 0x52574B31u;           // "RWK1": keep the walk's streams apart from
    const uint32_t k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(epoch >> 16) ^ 0x9E3779B9u;   // the resampler's
    hipLaunchKernelGGL(k_random_walk, dim3(grid_for((n + 1) / 2, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0,
                       (hipStream_t)stream, x, ldx, n, wa, z, ldz, k0, k1, (uint32_t)(epoch & 0xFFFFu));
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

int qsmc_tomo_canonicalize(qsmc_handle_t h, const double *basis, int32_t dim, double *x, int64_t ldx,
                           int64_t n, int32_t allow_subnormalized, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !basis || !x || n < 0) return QSMC_ERR_INVALID;
    if (n == 0) return QSMC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n, QSMC_BLOCK);
    switch (dim) {
        case 2:
            hipLaunchKernelGGL((k_tomo_canon<2>), dim3(grid), dim3(QSMC_BLOCK), 0, s, basis, x, ldx, n,
                               allow_subnormalized);
            break;
        case 3:
            return QSMC_ERR_UNSUPPORTED;   // d = 9 fits QSMC_MAX_D but no config needs it yet
        case 4: {
            if (n >= (1ll << 32)) return QSMC_ERR_UNSUPPORTED;
            int rc = ensure_iscratch(h, ((size_t)n + 4) * sizeof(unsigned int));
            if (rc) return rc;
            unsigned int *count = h->iscratch;              // [0] = list length; the list starts at [4]
            unsigned int *list = h->iscratch + 4;
            HIP_TRY(h, hipMemsetAsync(count, 0, sizeof(unsigned int), s));
            hipLaunchKernelGGL((k_tomo_classify<4>), dim3(grid), dim3(QSMC_BLOCK), 0, s, basis, x, ldx, n,
                               allow_subnormalized, list, count);
            hipLaunchKernelGGL((k_tomo_canon_list<4>), dim3(grid), dim3(QSMC_BLOCK), 0, s, basis, x, ldx,
                               allow_subnormalized, list, count);
            break;
        }
        default: return QSMC_ERR_UNSUPPORTED;
    }
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

// ---- posterior read-outs: sort by weight / by location, search a sorted table -------
__global__ __launch_bounds__(QSMC_BLOCK) void k_iota(int64_t *__restrict__ v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * QSMC_BLOCK) v[i] = i;
}

// out[k] = number of entries of the non-decreasing a[0..n) that are < q[k] (side 0, 'left') or <= q[k] (side 1)
__global__ __launch_bounds__(QSMC_BLOCK) void k_searchsorted(const double *__restrict__ a, int64_t n,
                                                             const double *__restrict__ q, int64_t m, int side,
                                                             int64_t *__restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; k < m; k += (int64_t)gridDim.x * QSMC_BLOCK) {
        const double v = q[k];
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            const bool go_right = side ? (a[mid] <= v) : (a[mid] < v);
            if (go_right) lo = mid + 1; else hi = mid;
        }
        out[k] = lo;
    }
}

int qsmc_argsort(qsmc_handle_t h, const double *keys, int64_t n, int32_t descending, double *keys_out,
                 int64_t *idx_out, qsmc_stream_t stream) {
    if (!h || !keys || !keys_out || !idx_out || n <= 0 || n >= (1ll << 31)) return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    size_t tmp_bytes = 0;
    // (size query: null temporary storage)
    hipError_t e = descending
        ? rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, keys, keys_out, (const int64_t *)nullptr, idx_out, (size_t)n, 0, 64, s)
        : rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_out, (const int64_t *)nullptr, idx_out, (size_t)n, 0, 64, s);
    HIP_TRY(h, e);
    const size_t iota_bytes = (size_t)n * sizeof(int64_t);
    const size_t need = iota_bytes + tmp_bytes + 256;
    if (h->sort_tmp_cap < need) {
        if (h->sort_tmp) HIP_TRY(h, hipFree(h->sort_tmp));
        h->sort_tmp = nullptr;
        h->sort_tmp_cap = 0;
        HIP_TRY(h, hipMalloc(&h->sort_tmp, need));
        h->sort_tmp_cap = need;
    }
    int64_t *iota = static_cast<int64_t *>(h->sort_tmp);
    void *tmp = static_cast<char *>(h->sort_tmp) + ((iota_bytes + 255) & ~(size_t)255);
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, s, iota, n);
    e = descending
        ? rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, keys, keys_out, (const int64_t *)iota, idx_out, (size_t)n, 0, 64, s)
        : rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_out, (const int64_t *)iota, idx_out, (size_t)n, 0, 64, s);
    HIP_TRY(h, e);
    return QSMC_OK;
}

int qsmc_searchsorted(qsmc_handle_t h, const double *a, int64_t n, const double *q, int64_t m, int32_t side,
                      int64_t *out, qsmc_stream_t stream) {
    if (!h || !a || !q || !out || n < 0 || m < 0 || (side != 0 && side != 1)) return QSMC_ERR_INVALID;
    if (m == 0) return QSMC_OK;
    hipLaunchKernelGGL(k_searchsorted, dim3(grid_for(m, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, (hipStream_t)stream, a, n,
                       q, m, side, out);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

// ---- host: all-gather of a few doubles between the ranks of one host through shared memory -----------------
// The per-datum collective of the sharded updater: layout and protocol of parallel.HostExchange
// (two banks by call parity; slot = [seq: int64 on its own 64-byte line][payload: max_len doubles]); this is
// its write / spin / read in C, no Python between the stores and the loads.
int qsmc_host_allgather(void *segment, int32_t rank, int32_t world, int32_t max_len, uint64_t k, const double *vec,
                        int32_t n, double *rows_out, double timeout_s) {
    if (!segment || !vec || !rows_out || rank < 0 || rank >= world || n < 0 || n > max_len) return QSMC_ERR_INVALID;
    constexpr int SEQ_STRIDE = 8;                              // int64s per sequence word (one cache line)
    volatile int64_t *seq = static_cast<volatile int64_t *>(segment);
    double *pay = reinterpret_cast<double *>(static_cast<char *>(segment) + (size_t)2 * world * SEQ_STRIDE * sizeof(int64_t));
    const int bank = (int)(k & 1u);
    double *mine = pay + ((size_t)bank * world + rank) * max_len;
    memcpy(mine, vec, (size_t)n * sizeof(double));
    std::atomic_thread_fence(std::memory_order_release);
    seq[((size_t)bank * world + rank) * SEQ_STRIDE] = (int64_t)k;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < world; ++r) {
        volatile int64_t *s = seq + ((size_t)bank * world + r) * SEQ_STRIDE;
        for (unsigned spins = 0; *s < (int64_t)k; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 0xffff) == 0xffff &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                return QSMC_ERR_UNSUPPORTED;                   // a peer did not arrive
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int r = 0; r < world; ++r)
        memcpy(rows_out + (size_t)r * n, pay + ((size_t)bank * world + r) * max_len, (size_t)n * sizeof(double));
    return QSMC_OK;
}

// The same exchange, reduced: tot_out[j] = rows[0][j] + rows[1][j] + ... in rank order (so every rank gets the
// same bits), except entry min_index (if >= 0), which is the minimum over ranks.  One call per datum of the
// sharded updater: normaliser, sum of squares, weight minimum, bad count and the fused moment sums together.
int qsmc_host_allreduce(void *segment, int32_t rank, int32_t world, int32_t max_len, uint64_t k, const double *vec,
                        int32_t n, int32_t min_index, double *rows_out, double *tot_out, double timeout_s) {
    if (!tot_out) return QSMC_ERR_INVALID;
    const int rc = qsmc_host_allgather(segment, rank, world, max_len, k, vec, n, rows_out, timeout_s);
    if (rc != QSMC_OK) return rc;
    for (int j = 0; j < n; ++j) {
        double acc = rows_out[j];
        if (j == min_index) {
            for (int r = 1; r < world; ++r) {
                const double v = rows_out[(size_t)r * n + j];
                acc = (v < acc || v != v) ? v : acc;            // a NaN weight minimum must reach every rank's guard
            }
        } else {
            for (int r = 1; r < world; ++r) acc += rows_out[(size_t)r * n + j];
        }
        tot_out[j] = acc;
    }
    return QSMC_OK;
}

// ---- host: sqrtm_psd by cyclic Jacobi (utils.py:593-607) --------------------------------------
int qsmc_sqrtm_psd(const double *A, int32_t d, double scale, double *S_out, double *err_out) {
    if (!A || !S_out || d < 1 || d > 64) return QSMC_ERR_INVALID;
    const int n = d;
    double *a = (double *)malloc(sizeof(double) * n * n * 3);
    if (!a) return QSMC_ERR_ALLOC;
    double *v = a + n * n, *sq = v + n * n;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            a[i * n + j] = 0.5 * (A[i * n + j] + A[j * n + i]);   // eigh reads one triangle; symmetrise
            v[i * n + j] = (i == j) ? 1.0 : 0.0;
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
    // S = V sqrt(max(lambda, 0)) V^T
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) {
                const double lam = a[k * n + k];
                const double r = lam <= 0.0 ? 0.0 : sqrt(lam);
                s += v[i * n + k] * r * v[j * n + k];
            }
            sq[i * n + j] = s;
        }
    if (err_out) {
        double e2 = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += sq[i * n + k] * sq[k * n + j];
                const double dlt = s - A[i * n + j];
                e2 += dlt * dlt;
            }
        *err_out = sqrt(e2);
    }
    for (int k = 0; k < n * n; ++k) S_out[k] = scale * sq[k];
    free(a);
    return QSMC_OK;
}

}  // extern "C"
