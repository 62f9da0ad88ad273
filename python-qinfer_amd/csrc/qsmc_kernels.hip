// qsmc_kernels.hip -- gfx950 kernels + C ABI (include/qsmc.h) for the SMC hot path.
//
// One translation unit: this file holds the handle (qsmc_ctx), the host-side helpers and the C ABI; the kernels
// are in kernels/*.hpp, included below in dependency order (update -> likelihood_moments -> scan -> resample ->
// walk_tomo), and qsmc_device.h has the per-particle device arithmetic (models, Philox, Box-Muller).
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

#include <dlfcn.h>

#include <rccl/rccl.h>              // types and prototypes only: the entry points are bound with dlsym (qsmc_comm_init)

#include "qsmc_device.h"

using namespace qsmc;

// =============================================================================================
// context
// =============================================================================================
struct LWDev;             // kernels/sqrtm.hpp
struct LWWide;            // kernels/wide.hpp
constexpr int WIDE_RING = 8;
struct Chain2Queue;       // passes of the design kernel in flight (qsmc_hypothetical_sums_begin / _collect)
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
    double *mapped_big;            // pinned host block for the d > 4 moments of a queued resample (MFMA_MOM_K doubles) ...
    double *mapped_big_dev;        // ... and its device alias
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
    unsigned int *tickets;         // device: arrival word of k_sum_columns_publish (self-resetting)
    int cu_count;                  // compute units this process can run on (bounds the resident grids of the barrier kernels)
    int cu_reported;               // what the device attribute says
    void *sort_tmp;                // qsmc_argsort: the temporary (key image, index) pair + digit histograms (kernels/sort.hpp)
    size_t sort_tmp_cap;           // in bytes
    double *tile_sums;             // sum of w' per update-kernel tile, written by the last qsmc_update_fused
    size_t tile_sums_cap;
    double *tile_prefix;           // monotone prefix of the unnormalised chunk sums (k_reduce_partials_scan) ...
    size_t tile_prefix_cap;
    unsigned long long tile_prefix_gen;   // ... of update number ts.gen (0: none)
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
    struct {                       // the prefix queued speculatively behind every update (qsmc_lw_arm_prefix)
        int enabled;
        double thresh;             // resample when (sum w')^2 / sum w'^2 < thresh
        int64_t n_out;
        uint64_t seed, epoch;
        int *gate;                 // device word the reducing kernel sets, the count kernel reads
        int launched;              // a speculative prefix followed update number `gen` ...
        unsigned long long gen;
        const double *w;           // ... of these weights
        int64_t n_in;
        int prof_slot;             // profiling-ring entry of that launch (-1: not timed)
        long long n_queued, n_adopted;   // speculative launches so far / resamples that found theirs done
    } spec;
    struct {                       // qsmc_lw_fuse_canonicalize: consumed by the next qsmc_lw_resample_philox
        int kind;                  // 0: none, 1: 2-qubit Pauli basis, 2: dense basis
        int allow_sub;
        const double *basis;
    } canon_next;
    struct {                       // the resample qsmc_step queued itself (adopted by a matching qsmc_lw_resample_philox)
        int valid;
        int canon_kind, canon_allow_sub;
        const double *canon_basis;
        double expect;
        qsmc_model_t model;
        int32_t postselect, d, maxiter;
        const double *x_in, *w;
        int64_t ldx_in, n_in, n_out, ldx_out;
        double norm, a;
        double mean[QSMC_MAX_D], S[QSMC_MAX_D * QSMC_MAX_D];
        uint64_t seed, epoch;
        double *x_out;
        hipStream_t stream;
        long long n_queued, n_adopted;
    } rsq;
    struct {                // proposal bank of the ordered sampler (kernels/resample.hpp): device buffers, grown on demand
        double *entries;
        long long capacity;           // in entries
        unsigned char *aux;           // per-item arrays, prefixes, counters, round lists
        size_t aux_cap;               // in bytes
        long long n_banked, n_rounds_leftover;   // resamples that used the bank (diagnostic)
    } bank;
    double expect_next;     // qsmc_lw_expect_redraws: consumed by the next qsmc_lw_resample_philox
    Chain2Queue *hypq;      // design passes queued by qsmc_hypothetical_sums_begin, waiting for _collect (allocated on first use)
    LWDev *lw_dev;          // device: Liu-West arguments of a d = 16 resample formed on the device (kernels/sqrtm.hpp)
    long long n_sqrt_dev, n_sqrt_agreed;   // square roots formed on the device by qsmc_step / of those, adopted after the host's check
    LWWide *lw_wide;        // device: a, mean, S of a d > 16 resample (kernels/wide.hpp), copied from ...
    LWWide *lw_wide_host;   // ... pinned host slots, WIDE_RING of them used in turn; a slot is rewritten only after the
    hipEvent_t lw_wide_ev[WIDE_RING];   // event recorded behind its copy has completed
    int lw_wide_next;
    double *wide_rho;       // device: the wide canonicalize's packed rho of every particle + its two E x E maps
    size_t wide_rho_cap;    // in doubles
    unsigned int *anc16;    // device: ancestors + canonicalize list of the split d = 16 sampler
    size_t anc16_cap;       // in bytes
    unsigned int *iscratch; // device integer scratch for the bucketed resampler
    size_t iscratch_cap;    // in bytes
    double *cdf_scratch;    // device CDF, materialised only for the direct sampler / global redraws
    size_t cdf_cap;         // in doubles
    struct {                       // RCCL communicator of the sharded updater (qsmc_comm_init); entry points bound at run time
        void *lib;
        ncclComm_t comm;
        int rank, nranks;
        double *buf;               // device: every rank's vector, [QSMC_MAX_RANKS][REDUCE_OUT_MAX]
        decltype(&ncclCommInitRank) CommInitRank;
        decltype(&ncclCommDestroy) CommDestroy;
        decltype(&ncclCommCount) CommCount;
        decltype(&ncclCommUserRank) CommUserRank;
        decltype(&ncclAllGather) AllGather;
        decltype(&ncclGetErrorString) GetErrorString;
    } cc;
    int profiling;
    hipEvent_t *prof_ev;   // QSMC_PROF_CAP (start, stop) pairs, created on first qsmc_set_profiling(1)
    int prof_n;            // profiled launches since the last qsmc_profile_read / set_profiling
    unsigned char *prof_tag;   // which kernel each ring entry timed (QSMC_PROF_*)
    int prof_stride;           // time every prof_stride-th launch of a tag (events cost ~10 us of queue drain each)
    unsigned prof_mask;        // tags that are timed at all (qsmc_set_profiling_tags; 0 = all)
    unsigned prof_seen[QSMC_PROF_NTAGS];     // launches seen per tag
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
constexpr int TICKET_WORDS = 16;                    // arrival words (k_sum_columns_publish uses the last one)
constexpr int QSMC_PROF_CAP = 4096;

// Test hooks (qsmc_test_hook; process-wide, all off by default).  Each selects an independent form of a kernel that the
// parity tests compare the default form against, or makes a rare branch common; none changes a result's law.
static bool g_multi_generic = false;      // k_update_multi: every tile through the general path
static bool g_redraw_no_small = false;    // k_bucket_redraw: the global-CDF form also for short queues
static bool g_hyp_no_chain = false;       // design passes of binomial experiments: thread-per-particle k_hyp_sums, not the walk
static bool g_canon_wide_jacobi = false;  // dim 5 .. 8 canonicalize: the one-lane eigenvector form for the listed particles
static bool g_tomo_dense = false;         // tomography updates read all d rows also for sparse measurement vectors
static double g_poisson_margin = 5.0;     // kappa in lambda = n_out - kappa sqrt(n_out) of k_bucket_counts
   // k_bucket_kick16: a workgroup takes consecutive slots of ONE part (the rounds 3-5 form)

// Census of the compute units this process can actually run on.  hipDeviceAttributeMultiprocessorCount reports the
// device's CUs; under a CU mask (HSA_CU_MASK / ROC_GLOBAL_CU_MASK) or other restrictions fewer are usable, and the
// resampler's barrier kernels size their grids by residency: a grid that cannot be resident as a whole would never
// pass its barrier (it aborts after a bounded spin, but aborts).  Every workgroup of a short saturating launch records
// (XCC, SE, SH, CU) of where it ran in a bitmap; the host counts the bits.
__global__ void k_cu_census(unsigned int *__restrict__ bitmap) {
    if (threadIdx.x == 0) {
        const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, all 32 bits
        const unsigned int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID[3:0]
        const unsigned int key = ((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu);                // CU_ID 11:8, SH_ID 12, SE_ID 15:13
        atomicOr(&bitmap[key >> 5], 1u << (key & 31u));
    }
    for (int i = 0; i < 12; ++i) __builtin_amdgcn_s_sleep(127);        // stay a few microseconds: later blocks go elsewhere
}

static int cu_census(int reported) {
    unsigned int *bm = nullptr;
    constexpr int WORDS = 4096 / 32;
    if (hipMalloc(&bm, WORDS * sizeof(unsigned int)) != hipSuccess) return reported;
    int usable = reported;
    unsigned int host[WORDS];
    if (hipMemset(bm, 0, sizeof(host)) == hipSuccess) {
        hipLaunchKernelGGL(k_cu_census, dim3(reported * 16), dim3(256), 0, 0, bm);
        if (hipMemcpy(host, bm, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess) {
            int n = 0;
            for (int w = 0; w < WORDS; ++w) n += __builtin_popcount(host[w]);
            if (n >= 1 && n <= reported) usable = n;
        }
    }
    (void)hipFree(bm);
    return usable;
}

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

#include "kernels/update.hpp"
#include "kernels/likelihood_moments.hpp"
#include "kernels/sqrtm.hpp"
#include "kernels/scan.hpp"
#include "kernels/resample.hpp"
#include "kernels/walk_tomo.hpp"
#include "kernels/wide.hpp"
#include "kernels/sort.hpp"
#include "kernels/user_jit.hpp"

// out[0..K) (device) -> pinned host block, then the completion word: the d > 4 moments of a resample queued by qsmc_step
__global__ void k_publish_big(const double *__restrict__ src, int K, double *__restrict__ mapped, unsigned long long *flag,
                              unsigned long long seq) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) mapped[k] = src[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        *reinterpret_cast<volatile unsigned long long *>(flag) = seq;
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
    if (model->kind == QSMC_MODEL_TOMOGRAPHY)
        for (int i = 0; i < model->d && i < QSMC_MAX_D; ++i)
            if (!(ep->meas[i] == 0.0)) out->nz_idx[out->nnz++] = i;          // (a NaN entry counts as present)
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

// d > QSMC_MAX_D (tomography): the measurement vector's nonzero entries from the host array the experiment points at
static int make_wide_args(const qsmc_model_t *model, const qsmc_expparam_t *ep, TomoWideArgs *out) {
    if (model->kind != QSMC_MODEL_TOMOGRAPHY || model->d <= QSMC_MAX_D || model->d > QSMC_MAX_D_WIDE) return QSMC_ERR_UNSUPPORTED;
    if (!ep->meas_wide) return QSMC_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    out->d = model->d;
    out->lik_pow = (model->likelihood_power == 1.0) ? 0.0 : model->likelihood_power;
    for (int i = 0; i < model->d; ++i)
        if (!(ep->meas_wide[i] == 0.0)) {                    // (a NaN entry counts as present)
            out->idx[out->nnz] = i;
            out->val[out->nnz++] = ep->meas_wide[i];
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
        case QSMC_MODEL_TOMOGRAPHY: return (m->d >= 1 && m->d <= QSMC_MAX_D_WIDE) ? QSMC_OK : QSMC_ERR_INVALID;
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
    ro.tile_prefix = nullptr;
    ro.tp_chunks = ro.tp_tpc = 0;
    ro.tp_ntiles = 0;
    ro.prefix_gate = nullptr;
    ro.prefix_thresh = 0.0;
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
    if (ro.tile_prefix) {                  // an update's reduction with the chunk-sum prefix beside it (ns = 3 + d + d (d + 1) / 2, d <= 4)
        switch (ns) {
#define LRS(N)                                                                                        \
    case N:                                                                                           \
        hipLaunchKernelGGL((k_reduce_partials_scan<N>), dim3(2), dim3(QSMC_BLOCK), 0, s, grid, ro);   \
        break;
            LRS(3) LRS(5) LRS(8) LRS(12) LRS(17) LRS(24) LRS(26) LRS(29) LRS(33) LRS(38)
#undef LRS
            default: return QSMC_ERR_INVALID;
        }
        HIP_TRY(h, hipGetLastError());
        return QSMC_OK;
    }
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
    if (h->spec.prof_slot >= 0) {          // the timed count launch behind this update left at its gate: not a count
        if (h->spec.launched && h->mapped[REDUCE_OUT_MAX - 3] != 1.0) h->prof_tag[h->spec.prof_slot] = QSMC_PROF_COUNTS_SKIPPED;
        h->spec.prof_slot = -1;
    }
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
    if (h->prof_mask && !(h->prof_mask & (1u << (tag & (QSMC_PROF_NTAGS - 1))))) return;
    if (h->prof_seen[tag & (QSMC_PROF_NTAGS - 1)]++ % (unsigned)h->prof_stride != 0) return;
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
    // streaming (non-temporal) hints once the pass no longer fits the 256 MB Infinity Cache: measured at N = 3e7 / 1e8
    // (d = 1) 128 -> 121 us / 455 -> 420 us; inside the cache they cost ~4 % (41.0 -> 42.5 us at N = 1e7), so not there
    const int nt = ((double)n * (double)(16 + 8 * e.d) > 3.0e8) ? 1 : 0;
#define LU(V, O)                                                                                          \
    do {                                                                                                  \
        if (e.lik_pow != 0.0)                                                                             \
            hipExtLaunchKernelGGL((k_update_fused<KIND, V, O, true>), dim3(grid), dim3(QSMC_BLOCK), 0, s, e0, e1, 0, \
                                  x, ldx, n, w_in, w_out, prev_norm, e, outcome, ro, nt);                 \
        else                                                                                              \
            hipExtLaunchKernelGGL((k_update_fused<KIND, V, O, false>), dim3(grid), dim3(QSMC_BLOCK), 0, s, e0, e1, 0, \
                                  x, ldx, n, w_in, w_out, prev_norm, e, outcome, ro, nt);                 \
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
    static_assert(NS + 1 <= REDUCE_OUT_MAX - 4, "reduce buffers too small");
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
    hipEvent_t he0 = nullptr, he1 = nullptr;
    prof_events(h, QSMC_PROF_HYP_SUMS, &he0, &he1);
    hipExtLaunchKernelGGL((k_hyp_sums<KIND, NO>), dim3(grid), dim3(QSMC_BLOCK), 0, s, he0, he1, 0, x, ldx, n, w, norm, ha, ro);
    HIP_TRY(h, hipGetLastError());
    rc = launch_reduce(h, NS, grid, ro, s);
    if (rc) return rc;
    rc = wait_reduction(h, s);
    if (rc) return rc;
    memcpy(out_host, h->mapped, (size_t)n_o * PER * sizeof(double));
    return QSMC_OK;
}

// binomial models, consecutive outcomes: a lane per particle walks the outcomes of a pass from both ends (k_hyp_sums_chain2);
// the binomial coefficients and the ln C term of sum w L ln L are applied on the host, to the finished sums.
// Two halves, so that the passes of SEVERAL experiments queue back to back and the host waits once (bayes_risk over a
// design: per experiment ~15 us of launch + publish + completion-word round trip and the caller's own per-call work
// otherwise sit between the kernels): chain2_enqueue launches a pass and publishes its sums to a slot of the pinned
// block; chain2_collect, after the wait, turns a slot into the caller's rows.
constexpr int CHAIN2_MAX_SLOTS = 2 * 13;
constexpr int CHAIN2_PENDING_MAX = 12;             // passes in flight: 12 x 78 doubles <= the pinned block's 1024
struct Chain2Pending {
    int off;                       // first double of the pass's sums in h->mapped_big
    int n_o, n_up, nh, per, what, d_out;
    double nm;
    int64_t outcomes[CHAIN2_MAX_SLOTS];
    double lc[CHAIN2_MAX_SLOTS];
    double *out;                   // the pass's rows of the caller's array
};

template <int KIND, int WHAT, int NH>
static int chain2_enqueue(qsmc_ctx *h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                          const double *w, double norm, const qsmc_expparam_t *exp, const int64_t *outcomes, int n_o,
                          const double *shift, double *out_host, int off, Chain2Pending *pend, hipStream_t s) {
    constexpr bool LOG = (WHAT & HYP_WHAT_LOG) != 0, MOM = (WHAT & HYP_WHAT_MOM) != 0;
    constexpr int DO = Model<KIND>::D <= 4 ? Model<KIND>::D : 0;       // the caller's row: [N, SL, S1[DO], S2[DO]]
    constexpr int D = MOM ? DO : 0;
    constexpr int PER = 1 + (LOG ? 1 : 0) + 2 * D;
    constexpr int NS = 2 * NH * PER;
    static_assert(NS <= 512 && 2 * NH <= CHAIN2_MAX_SLOTS, "the pinned block of the wide sums holds 512 doubles a pass");
    if (n_o < 1 || n_o > 2 * NH || off < 0 || off + NS > SQRT_MAPPED_DOUBLES) return QSMC_ERR_INVALID;
    // one resident round: the NS wave reductions at a workgroup's end are worth a few particles each, and a grid of
    // 2048 workgroups on 512 - 768 resident slots ends in a ragged last round
    static const int per_cu = [] {
        int b = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_hyp_sums_chain2<KIND, WHAT, NH>, QSMC_BLOCK, 0);
        if (e != hipSuccess || b < 1) b = 2;
        return b > 8 ? 8 : b;
    }();
    int grid = grid_for(n, QSMC_BLOCK * 4);
    if (grid > per_cu * h->cu_count) grid = per_cu * h->cu_count;
    int rc = ensure_partials(h, (size_t)grid * (NS + 1));
    if (rc) return rc;
    rc = ensure_scratch(h, 256 + 512);
    if (rc) return rc;
    Chain2Args ca;
    memset(&ca, 0, sizeof(ca));
    make_exp_args(model, exp, outcomes[0], &ca.base);
    ca.n_up = (n_o + 1) / 2;
    ca.n_dn = n_o - ca.n_up;
    ca.use_powi = exp->n_meas <= 64 ? 1 : 0;
    ca.k_first = (unsigned)outcomes[0];
    ca.k_last = (unsigned)outcomes[n_o - 1];
    pend->off = off;
    pend->n_o = n_o;
    pend->n_up = ca.n_up;
    pend->nh = NH;
    pend->per = PER;
    pend->what = WHAT;
    pend->d_out = DO;
    pend->nm = (double)exp->n_meas;
    pend->out = out_host;
    for (int j = 0; j < n_o; ++j) {
        ExpArgs tmp;
        make_exp_args(model, exp, outcomes[j], &tmp);
        pend->outcomes[j] = outcomes[j];
        pend->lc[j] = tmp.log_comb;
        if (j == 0) { ca.comb_first = tmp.comb; ca.lc_first = tmp.log_comb; }
        if (j == n_o - 1) { ca.comb_last = tmp.comb; ca.lc_last = tmp.log_comb; }
    }
    if (shift) for (int m = 0; m < model->d && m < QSMC_MAX_D; ++m) ca.shift[m] = shift[m];
    ReduceOut ro;
    memset(&ro, 0, sizeof(ro));
    ro.partials = h->partials;
    hipEvent_t he0 = nullptr, he1 = nullptr;
    prof_events(h, QSMC_PROF_HYP_SUMS, &he0, &he1);
    hipExtLaunchKernelGGL((k_hyp_sums_chain2<KIND, WHAT, NH>), dim3(grid), dim3(QSMC_BLOCK), 0, s, he0, he1, 0, x, ldx, n, w, norm, ca, ro);
    // column sums and their publish to the pinned block in ONE launch (round 5; k_sum_columns + k_publish_big before: two
    // launches, ~9 us per pass): every workgroup stores its columns, fences, draws a ticket; the last sets the word
    const unsigned long long seq = ++h->seq;           // (the completion word steps through the passes; the host waits for the last)
    hipLaunchKernelGGL(k_sum_columns_publish, dim3((NS + QSMC_WAVES_PER_BLOCK - 1) / QSMC_WAVES_PER_BLOCK), dim3(QSMC_BLOCK), 0, s,
                       h->partials, grid, NS, h->mapped_big_dev + off, h->flag_dev, seq, h->tickets + TICKET_WORDS - 2);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

// slot -> outcome, C(n, k) / C(n, k_start) by the ratios of neighbouring coefficients, ln C for the entropy term
static void chain2_collect(const qsmc_ctx *h, const Chain2Pending &pd) {
    const bool LOG = (pd.what & HYP_WHAT_LOG) != 0, MOM = (pd.what & HYP_WHAT_MOM) != 0;
    const int DO = pd.d_out, D = MOM ? DO : 0, PER_OUT = 2 + 2 * DO, B1 = 1 + (LOG ? 1 : 0);
    const double *sums = h->mapped_big + pd.off;
    double scale = 1.0;
    for (int j = 0; j < pd.n_o; ++j) {
        const bool up = j < pd.n_up;
        const int o = up ? j : pd.n_o - 1 - (j - pd.n_up);               // index into this pass's outcomes
        const int slot = up ? j : pd.nh + (j - pd.n_up);
        if (j == 0 || j == pd.n_up) scale = 1.0;
        const double *v = sums + (size_t)slot * pd.per;
        double *row = pd.out + (size_t)o * PER_OUT;
        for (int q = 0; q < PER_OUT; ++q) row[q] = NAN;                  // what the caller did not ask for
        row[0] = scale * v[0];
        if (LOG) row[1] = scale * v[1] + pd.lc[o] * row[0];
        for (int m = 0; m < D; ++m) {
            row[2 + m] = scale * v[B1 + m];
            row[2 + DO + m] = scale * v[B1 + D + m];
        }
        const double k = (double)pd.outcomes[o];
        scale *= up ? (pd.nm - k) / (k + 1.0) : k / (pd.nm - k + 1.0);   // to the next slot of this walk
    }
}

struct Chain2Queue {
    Chain2Pending pend[CHAIN2_PENDING_MAX];
    int count = 0, off = 0;
};

static int chain2_flush(qsmc_ctx *h, Chain2Queue &q, hipStream_t s) {
    if (q.count == 0) return QSMC_OK;
    const int rc = wait_reduction(h, s);               // (the last pass queued holds h->seq)
    // (the queue lives on the handle: also on the error path nothing may stay in it -- its entries hold the CALLER's row
    //  pointers, which a caller that got an error back is free to release)
    if (rc == QSMC_OK)
        for (int i = 0; i < q.count; ++i) chain2_collect(h, q.pend[i]);
    q.count = 0;
    q.off = 0;
    return rc;
}

template <int KIND, int WHAT, int NH>
static int chain2_go(qsmc_ctx *h, Chain2Queue &q, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                     const double *w, double norm, const qsmc_expparam_t *exp, const int64_t *outcomes, int n_o,
                     const double *shift, double *out_host, hipStream_t s) {
    constexpr int NS = 2 * chain2_lane_sums<KIND, WHAT, NH>();
    if (q.count == CHAIN2_PENDING_MAX || q.off + NS > SQRT_MAPPED_DOUBLES) {
        const int rc = chain2_flush(h, q, s);
        if (rc) return rc;
    }
    const int rc = chain2_enqueue<KIND, WHAT, NH>(h, model, x, ldx, n, w, norm, exp, outcomes, n_o, shift, out_host, q.off,
                                                  &q.pend[q.count], s);
    if (rc) return rc;
    q.count += 1;
    q.off += NS;
    return QSMC_OK;
}

// One experiment.  Passes of the two-ended walk are left in `q` (the caller flushes it); every other path flushes `q`
// first and returns with its rows filled.
template <int KIND>
static int hyp_dispatch(qsmc_ctx *h, Chain2Queue &q, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                        const double *w, double norm, const qsmc_expparam_t *exp, const int64_t *outcomes,
                        int n_o, const double *shift, double *out_host, hipStream_t s, int what) {
    // Outcome lists that are not a binomial experiment's consecutive domain go through the thread-per-particle kernel
    // (k_hyp_sums) in groups of eight: a pass re-reads x and w (16 + 8 d bytes per particle) and the running sums (PER per
    // outcome) live in registers.  (Rounds 3-5 also kept a 32-outcome form of it and a lane-per-outcome kernel; both lost to
    // the two-ended walk on every binomial design and were removed in round 6.)
    constexpr int D = Model<KIND>::D <= 4 ? Model<KIND>::D : 0;
    constexpr int PER = 2 + 2 * D;
    int done = 0;
    constexpr bool BINOMIAL = KIND == QSMC_MODEL_BINOMIAL_PRECESSION || KIND == QSMC_MODEL_BINOMIAL_RB ||
                              KIND == QSMC_MODEL_BINOMIAL_RB_INTERLEAVED;
    const bool no_chain = g_hyp_no_chain;                                           // (test hook: the independent form)
    // consecutive outcomes inside [0, n_meas] (the domain of a binomial experiment, in order): passes of equal size
    bool consecutive = BINOMIAL && !no_chain && n_o > 2 && model->likelihood_power == 0.0 && outcomes[0] >= 0 &&
                       (uint64_t)outcomes[n_o - 1] <= exp->n_meas;
    for (int o = 1; o < n_o && consecutive; ++o) consecutive = outcomes[o] == outcomes[0] + o;
    what &= HYP_WHAT_LOG | HYP_WHAT_MOM;
    if (what == 0 || D == 0) what |= HYP_WHAT_LOG;
    // slots per pass of the two-ended walk: 2 NH, NH from the registers the sums take (2 NH x PER doubles: ~56 for three
    // waves per SIMD; the moments at D = 1 take 78 -- a 26-outcome experiment in ONE pass at two waves per SIMD, 138 us,
    // against two passes of 13 at three waves, 216 us)
    // slots per pass of the two-ended walk: 2 NH, NH from the registers the sums of ONE direction take in a lane (NH x PER
    // doubles: <= 39 for four waves per SIMD, <= 58 for three)
    constexpr int NH_LOG = 13, NH_MOM = D <= 1 ? 13 : (D <= 3 ? 8 : 6), NH_ALL = D <= 1 ? 13 : (D <= 3 ? 7 : 5);
    const int chain_slots = 2 * (what == HYP_WHAT_LOG ? NH_LOG : (what == HYP_WHAT_MOM ? NH_MOM : NH_ALL));
    const int chain_passes = (n_o + chain_slots - 1) / chain_slots;
    const int chain_take = (n_o + chain_passes - 1) / chain_passes;
    while (done < n_o) {
        const int m = n_o - done;
        int take, rc;
        if (consecutive) {
            take = m < chain_take ? m : chain_take;
            if constexpr (BINOMIAL) {
                const int64_t *oc = outcomes + done;
                double *oh = out_host + (size_t)done * PER;
#define CG(W, N_) chain2_go<KIND, W, N_>(h, q, model, x, ldx, n, w, norm, exp, oc, take, shift, oh, s)
                if (what == HYP_WHAT_LOG) rc = CG(HYP_WHAT_LOG, NH_LOG);
                else if (what == HYP_WHAT_MOM) rc = CG(HYP_WHAT_MOM, NH_MOM);
                else rc = CG(HYP_WHAT_LOG | HYP_WHAT_MOM, NH_ALL);
#undef CG
            } else
                rc = QSMC_ERR_INVALID;
        } else if ((rc = chain2_flush(h, q, s)) != QSMC_OK) {
            return rc;
        } else if (m <= 2) {
            take = m;
            rc = hyp_launch<KIND, 2>(h, model, x, ldx, n, w, norm, exp, outcomes + done, take, shift,
                                     out_host + (size_t)done * PER, s);
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

// The list pass of a 2-qubit canonicalize: the eigenvector-free form (k_tomo_canon_list_fast, round 5) on the list, then
// the eigenvector form on what that one flagged (count[1] entries of list2: none on a Ginibre-like cloud -- the launch
// leaves at once).
template <class Basis>
static void launch_canon_list4(Basis B, int grid, hipStream_t s, hipEvent_t l0, hipEvent_t l1, double *x, int64_t ldx,
                               int32_t allow_sub, const unsigned int *list, unsigned int *count, unsigned int *list2) {
    hipExtLaunchKernelGGL((k_tomo_canon_list_fast<Basis>), dim3(grid), dim3(QSMC_BLOCK), 0, s, l0, l1, 0, B, x, ldx, allow_sub,
                          list, count, list2);
    hipLaunchKernelGGL((k_tomo_canon_list<4, Basis>), dim3(64), dim3(QSMC_BLOCK), 0, s, B, x, ldx, allow_sub, list2, count + 1);
}

template <class Basis>
static int canon_dim4(qsmc_ctx *h, Basis B, double *x, int64_t ldx, int64_t n, int32_t allow_subnormalized, hipStream_t s) {
    if (n >= (1ll << 32)) return QSMC_ERR_UNSUPPORTED;
    const int grid = grid_for(n, QSMC_BLOCK);
    int rc = ensure_iscratch(h, (2 * (size_t)n + 8) * sizeof(unsigned int));
    if (rc) return rc;
    unsigned int *count = h->iscratch;              // [0] = list length, [1] = length of the second list; the list starts at [4]
    unsigned int *list = h->iscratch + 4, *list2 = h->iscratch + 4 + n + 4;
    HIP_TRY(h, hipMemsetAsync(count, 0, 2 * sizeof(unsigned int), s));
    hipEvent_t c0 = nullptr, c1 = nullptr, l0 = nullptr, l1 = nullptr;
    prof_events(h, QSMC_PROF_CANON_CLASSIFY, &c0, &c1);
    prof_events(h, QSMC_PROF_CANON_LIST, &l0, &l1);
    const int cgrid = grid < 1024 ? grid : 1024;       // (few flushes per workgroup: see k_tomo_classify)
    hipExtLaunchKernelGGL((k_tomo_classify<4, Basis>), dim3(cgrid), dim3(QSMC_BLOCK), 0, s, c0, c1, 0, B, x, ldx, n,
                          allow_subnormalized, list, count);
    launch_canon_list4(B, grid, s, l0, l1, x, ldx, allow_subnormalized, list, count, list2);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" { static int ensure_anc16(qsmc_ctx *h, size_t bytes); }
// device scratch of the wide canonicalize: rho of every particle in packed Hermitian form (E x n doubles) + the two E x E maps
static int ensure_wide_rho(qsmc_ctx *h, size_t doubles) {
    if (h->wide_rho_cap >= doubles) return QSMC_OK;
    if (h->wide_rho) HIP_TRY(h, hipFree(h->wide_rho));
    h->wide_rho = nullptr;
    h->wide_rho_cap = 0;
    HIP_TRY(h, hipMalloc(&h->wide_rho, doubles * sizeof(double)));
    h->wide_rho_cap = doubles;
    return QSMC_OK;
}

// dim 5 .. 8 (kernels/wide.hpp): rho_packed = Mb x for every particle on the matrix cores, LDL^H pivot test (positive-definite
// particles are finished there), one-sided Jacobi for the listed rest, x = Me R_packed + the trace renormalisation
template <int DIM>
static int canon_wide(qsmc_ctx *h, const double *basis, double *x, int64_t ldx, int64_t n, int32_t allow_subnormalized,
                      hipStream_t s) {
    if (!basis) return QSMC_ERR_INVALID;                     // (dense contraction with the basis tensor, whatever the basis)
    if (n >= (1ll << 31)) return QSMC_ERR_UNSUPPORTED;       // (list entries: 31 bits of index + a flag)
    int rc = ensure_anc16(h, ((size_t)n + 16) * sizeof(unsigned int));
    if (rc) return rc;
    unsigned int *count = h->anc16, *list = h->anc16 + 4;
    HIP_TRY(h, hipMemsetAsync(count, 0, 4 * sizeof(unsigned int), s));
    hipEvent_t c0 = nullptr, c1 = nullptr, l0 = nullptr, l1 = nullptr;
    if constexpr (DIM == 8 || DIM == 5) {                    // (test hook: the independent, eigenvector form, built for dim 5 and 8)
        if (g_canon_wide_jacobi) {
            prof_events(h, QSMC_PROF_CANON_CLASSIFY, &c0, &c1);
            hipExtLaunchKernelGGL((k_tomo_classify_wide<DIM>), dim3(grid_for(n, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, s, c0, c1, 0,
                                  basis, x, ldx, n, allow_subnormalized, list, count);
            prof_events(h, QSMC_PROF_CANON_LIST, &l0, &l1);
            hipExtLaunchKernelGGL((k_tomo_canon_list_wide<DIM>), dim3(grid_for(n, 64)), dim3(64), 0, s, l0, l1, 0, basis, x, ldx,
                                  allow_subnormalized, list, count);
            HIP_TRY(h, hipGetLastError());
            return QSMC_OK;
        }
    }
    constexpr int E = DIM * DIM, NB = (E + 15) / 16, DP = 16 * NB;
    const int64_t ld = (n + 3) / 4 * 4;
    rc = ensure_wide_rho(h, (size_t)E * (size_t)ld + 2 * (size_t)DP * DP);
    if (rc) return rc;
    double *rho = h->wide_rho, *Mb = h->wide_rho + (size_t)E * (size_t)ld, *Me = Mb + (size_t)DP * DP;
    hipLaunchKernelGGL((k_canon_mats<DIM>), dim3((DP * DP + QSMC_BLOCK - 1) / QSMC_BLOCK), dim3(QSMC_BLOCK), 0, s, basis, Mb, Me);
    hipEvent_t b0 = nullptr, b1 = nullptr, e0 = nullptr, e1 = nullptr;
    prof_events(h, QSMC_PROF_CANON_BUILD, &b0, &b1);
    const unsigned ggrid = (unsigned)((n + GEMMW_PER_BLOCK - 1) / GEMMW_PER_BLOCK);
    hipExtLaunchKernelGGL((k_gemm_wide<NB, 0>), dim3(ggrid), dim3(GEMMW_BT), 0, s, b0, b1, 0, (const double *)Mb, E, x, ldx, n, rho,
                          ld, (const unsigned int *)nullptr, (const unsigned int *)nullptr, DIM, allow_subnormalized);
    prof_events(h, QSMC_PROF_CANON_CLASSIFY, &c0, &c1);
    hipExtLaunchKernelGGL((k_tomo_ldl_wide<DIM>), dim3(grid_for(n, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, s, c0, c1, 0,
                          (const double *)rho, ld, x, ldx, n, allow_subnormalized, list, count);
    prof_events(h, QSMC_PROF_CANON_LIST, &l0, &l1);
    hipExtLaunchKernelGGL((k_tomo_jacobi_wide<DIM>), dim3(grid_for(n, 64)), dim3(64), 0, s, l0, l1, 0, rho, ld, list,
                          (const unsigned int *)count);
    prof_events(h, QSMC_PROF_CANON_EXPAND, &e0, &e1);
    hipExtLaunchKernelGGL((k_gemm_wide<NB, 1>), dim3(ggrid), dim3(GEMMW_BT), 0, s, e0, e1, 0, (const double *)Me, E, x, ldx, n, rho,
                          ld, (const unsigned int *)list, (const unsigned int *)count, DIM, allow_subnormalized);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}


extern "C" {

int qsmc_abi_version(void) { return QSMC_ABI_VERSION; }

int qsmc_test_hook(int32_t hook, double value) {
    switch (hook) {
        case QSMC_HOOK_MULTI_GENERIC: g_multi_generic = value != 0.0; return QSMC_OK;
        case QSMC_HOOK_REDRAW_NO_SMALL: g_redraw_no_small = value != 0.0; return QSMC_OK;
        case QSMC_HOOK_HYP_NO_CHAIN: g_hyp_no_chain = value != 0.0; return QSMC_OK;
        case QSMC_HOOK_TOMO_DENSE: g_tomo_dense = value != 0.0; return QSMC_OK;
        case QSMC_HOOK_POISSON_MARGIN: g_poisson_margin = value; return QSMC_OK;
        case QSMC_HOOK_CANON_WIDE_JACOBI: g_canon_wide_jacobi = value != 0.0; return QSMC_OK;
        default: return QSMC_ERR_INVALID;
    }
}

const char *qsmc_strerror(int status) {
    switch (status) {
        case QSMC_OK: return "ok";
        case QSMC_ERR_INVALID: return "invalid argument";
        case QSMC_ERR_HIP: return "HIP runtime error";
        case QSMC_ERR_ALLOC: return "allocation failed";
        case QSMC_ERR_UNSUPPORTED: return "unsupported configuration";
        case QSMC_ERR_TIMEOUT: return "timed out waiting for a peer";
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
    if (e == hipSuccess) e = hipMalloc(&h->tickets, TICKET_WORDS * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(h->tickets, 0, TICKET_WORDS * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc(&h->spec.gate, sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->spec.gate, 0, sizeof(int));
    h->spec.prof_slot = -1;
    if (e == hipSuccess) e = hipMalloc(&h->red_out, REDUCE_OUT_MAX * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc(&h->mapped, REDUCE_OUT_MAX * sizeof(double), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&h->mapped_dev, h->mapped, 0);
    if (e == hipSuccess) e = hipHostMalloc(&h->mapped_big, SQRT_MAPPED_DOUBLES * sizeof(double), hipHostMallocMapped);
    if (e == hipSuccess) e = hipMalloc(&h->lw_dev, sizeof(LWDev));
    if (e == hipSuccess) e = hipMemset(h->lw_dev, 0, sizeof(LWDev));
    if (e == hipSuccess) e = hipMalloc(&h->lw_wide, sizeof(LWWide));
    if (e == hipSuccess) e = hipHostMalloc(&h->lw_wide_host, WIDE_RING * sizeof(LWWide), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&h->mapped_big_dev, h->mapped_big, 0);
    if (e == hipSuccess) e = hipHostMalloc(&h->flag, 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&h->flag_dev, h->flag, 0);
    if (e == hipSuccess) *h->flag = 0ull;
    if (e == hipSuccess) {
        int cus = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        h->cu_reported = cus > 0 ? cus : 1;
        h->cu_count = e == hipSuccess ? cu_census(h->cu_reported) : h->cu_reported;
    }
    if (e != hipSuccess) {
        delete h;
        return QSMC_ERR_HIP;
    }
    *out = h;
    return QSMC_OK;
}

int qsmc_destroy(qsmc_handle_t h) {
    if (!h) return QSMC_OK;
    (void)qsmc_comm_destroy(h);
    if (h->partials) (void)hipFree(h->partials);
    if (h->rs_offsets) (void)hipFree(h->rs_offsets);
    if (h->tile_sums) (void)hipFree(h->tile_sums);
    if (h->tile_prefix) (void)hipFree(h->tile_prefix);
    if (h->sort_tmp) (void)hipFree(h->sort_tmp);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->counter) (void)hipFree(h->counter);
    if (h->gbar) (void)hipFree(h->gbar);
    delete h->hypq;
    if (h->tickets) (void)hipFree(h->tickets);
    if (h->spec.gate) (void)hipFree(h->spec.gate);
    if (h->iscratch) (void)hipFree(h->iscratch);
    if (h->anc16) (void)hipFree(h->anc16);
    if (h->lw_dev) (void)hipFree(h->lw_dev);
    if (h->lw_wide) (void)hipFree(h->lw_wide);
    if (h->wide_rho) (void)hipFree(h->wide_rho);
    if (h->lw_wide_host) (void)hipHostFree(h->lw_wide_host);
    for (int i = 0; i < WIDE_RING; ++i)
        if (h->lw_wide_ev[i]) (void)hipEventDestroy(h->lw_wide_ev[i]);
    if (h->bank.entries) (void)hipFree(h->bank.entries);
    if (h->bank.aux) (void)hipFree(h->bank.aux);
    if (h->cdf_scratch) (void)hipFree(h->cdf_scratch);
    if (h->red_out) (void)hipFree(h->red_out);
    if (h->mapped) (void)hipHostFree(h->mapped);
    if (h->flag) (void)hipHostFree(h->flag);
    if (h->mapped_big) (void)hipHostFree(h->mapped_big);
    if (h->prof_ev) {
        for (int i = 0; i < 2 * QSMC_PROF_CAP; ++i) (void)hipEventDestroy(h->prof_ev[i]);
        free(h->prof_ev);
        free(h->prof_tag);
    }
    delete h;
    return QSMC_OK;
}

int qsmc_device_cus(qsmc_handle_t h, int32_t *usable_out, int32_t *reported_out) {
    if (!h) return QSMC_ERR_INVALID;
    if (usable_out) *usable_out = h->cu_count;
    if (reported_out) *reported_out = h->cu_reported;
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

int qsmc_set_profiling_tags(qsmc_handle_t h, uint32_t tag_mask) {
    if (!h) return QSMC_ERR_INVALID;
    h->prof_mask = tag_mask;
    return QSMC_OK;
}

int qsmc_last_update_kernel_ms(qsmc_handle_t h, float *ms_out) {
    if (!h || !ms_out || !h->prof_ev || h->prof_n < 1) return QSMC_ERR_INVALID;
    int slot = -1;
    for (int i = h->prof_n - 1; i >= 0 && i > h->prof_n - 1 - QSMC_PROF_CAP; --i)
        if (h->prof_tag[i % QSMC_PROF_CAP] == QSMC_PROF_UPDATE || h->prof_tag[i % QSMC_PROF_CAP] == QSMC_PROF_UPDATE_ONES) { slot = i % QSMC_PROF_CAP; break; }
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
    if (model->d > QSMC_MAX_D) {                             // three-qubit tomography and its like: kernels/wide.hpp
        for (int e = 0; e < n_e; ++e) {
            TomoWideArgs wa;
            rc = make_wide_args(model, &exps[e], &wa);
            if (rc) return rc;
            for (int o = 0; o < n_o; ++o)
                hipLaunchKernelGGL(k_likelihood_wide, dim3(grid), dim3(QSMC_BLOCK), 0, s, x, ldx, n, wa, outcomes[o],
                                   L_out + ((size_t)o * n_e + e) * (size_t)n);
        }
        HIP_TRY(h, hipGetLastError());
        return QSMC_OK;
    }
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

static int resample_prefix(qsmc_ctx *h, const double *w, int64_t n_in, double norm, int64_t n_out, uint64_t seed,
                           uint64_t epoch, hipStream_t s, bool speculative);

// the chunk-sum prefix beside the reduction (k_reduce_partials_scan), for the update kernels that leave tile sums
static bool reduce_scan_has(int ns) {
    return ns == 3 || ns == 5 || ns == 8 || ns == 12 || ns == 17 || ns == 24 || ns == 26 || ns == 29 || ns == 33 || ns == 38;
}
static int setup_tile_prefix(qsmc_ctx *h, ReduceOut &ro, int64_t n, int per_block, int ns) {
    const int64_t tp_chunks = (n + BUCKET_CHUNK - 1) / BUCKET_CHUNK;
    if (!(ro.tile_sums && tp_chunks <= TILE_PREFIX_MAX_CHUNKS && reduce_scan_has(ns))) return QSMC_OK;
    if (h->tile_prefix_cap < (size_t)tp_chunks + 1) {
        if (h->tile_prefix) HIP_TRY(h, hipFree(h->tile_prefix));
        h->tile_prefix = nullptr;
        h->tile_prefix_cap = 0;
        HIP_TRY(h, hipMalloc(&h->tile_prefix, (size_t)(TILE_PREFIX_MAX_CHUNKS + 1) * sizeof(double)));
        h->tile_prefix_cap = TILE_PREFIX_MAX_CHUNKS + 1;
    }
    ro.tile_prefix = h->tile_prefix;
    ro.tp_chunks = (int)tp_chunks;
    ro.tp_tpc = BUCKET_CHUNK / per_block * QSMC_WAVES_PER_BLOCK;
    ro.tp_ntiles = (long long)((n + per_block - 1) / per_block) * QSMC_WAVES_PER_BLOCK;
    h->tile_prefix_gen = h->ts.gen;
    return QSMC_OK;
}

int qsmc_update_fused(qsmc_handle_t h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                      const double *w_in, double *w_out, double prev_norm, const qsmc_expparam_t *exp,
                      int64_t outcome, double *stats_dev, qsmc_update_stats_t *stats_host, double *moments_host,
                      qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->rsq.valid = 0; h->canon_next.kind = 0; h->expect_next = 0.0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
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
    // (grids of 512 ... 2442 workgroups measured at N = 1e7: 2048 and 1024 tie, 814 and below lose 4 us, 2442 wins 0.7 us
    //  in this kernel and loses 0.85 us in the reducing one)
    const int grid = grid_for(n, per_block);
    rc = ensure_partials(h, (size_t)grid * (ns + 1));
    if (rc) return rc;
    const bool wide = d > QSMC_MAX_D;
    ExpArgs ea;
    TomoWideArgs wa;
    if (wide) {
        rc = make_wide_args(model, exp, &wa);
        if (rc) return rc;
    } else {
        make_exp_args(model, exp, outcome, &ea);
    }
    ReduceOut ro = make_reduce(h, stats_host || moments_host, stats_dev);
    // per-tile sums of the new weights: a resample that follows this update takes its chunk sums from them
    if (BUCKET_CHUNK % per_block == 0 &&
        ensure_tile_sums(h, (size_t)((n + BUCKET_CHUNK - 1) / BUCKET_CHUNK) * (BUCKET_CHUNK / per_block) * QSMC_WAVES_PER_BLOCK) == QSMC_OK) {   // (whole chunks: the update kernel zero-fills the last one)
        ro.tile_sums = h->tile_sums;
        h->ts.w = w_out;
        h->ts.n = n;
        h->ts.tile = per_block;
    } else {
        h->ts.w = nullptr;
    }
    ++h->ts.gen;
    h->ts.armed = 0;
    h->spec.launched = 0;
    rc = setup_tile_prefix(h, ro, n, per_block, ns);
    if (rc) return rc;
    if (h->spec.enabled && ro.tile_sums && ro.failed_dst) {
        // the resampler's weight-only prefix goes out right behind the reduction, gated on the device-side ESS test
        ro.prefix_gate = h->spec.gate;
        ro.prefix_thresh = h->spec.thresh;
    }
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
        case QSMC_MODEL_TOMOGRAPHY: {
            if (wide) {
                // 16 < d <= 64: the rows the measurement vector touches, streamed (kernels/wide.hpp)
                hipEvent_t e0 = nullptr, e1 = nullptr;
                prof_events(h, w_in ? QSMC_PROF_UPDATE : QSMC_PROF_UPDATE_ONES, &e0, &e1);
                const bool nt_w = (double)n * (double)(16 + 8 * wa.nnz) > 3.0e8;
#define LW_(V, O)                                                                                                          \
    do {                                                                                                                   \
        if (nt_w)                                                                                                          \
            hipExtLaunchKernelGGL((k_update_tomo_wide<V, O, true>), dim3(grid), dim3(QSMC_BLOCK), 0, s, e0, e1, 0, x, ldx, n, \
                                  w_in, w_out, prev_norm, wa, outcome, ro);                                                \
        else                                                                                                               \
            hipExtLaunchKernelGGL((k_update_tomo_wide<V, O, false>), dim3(grid), dim3(QSMC_BLOCK), 0, s, e0, e1, 0, x, ldx, n, \
                                  w_in, w_out, prev_norm, wa, outcome, ro);                                                \
    } while (0)
                if (vec2 && w_in) LW_(2, false);
                else if (vec2) LW_(2, true);
                else if (w_in) LW_(1, false);
                else LW_(1, true);
#undef LW_
                break;
            }
            // a measurement vector with at most four nonzero entries (a Pauli measurement has two): read those rows only
            const bool dense_env = g_tomo_dense;                  // (test hook: the dense form, for the same-bits test)
            if (!dense_env && vec2 && ea.lik_pow == 0.0 && ea.nnz >= 1 && ea.nnz <= 4) {
                hipEvent_t e0 = nullptr, e1 = nullptr;
                prof_events(h, w_in ? QSMC_PROF_UPDATE : QSMC_PROF_UPDATE_ONES, &e0, &e1);
#define LT(NZ)                                                                                                          \
    case NZ:                                                                                                            \
        if (w_in)                                                                                                       \
            hipExtLaunchKernelGGL((k_update_tomo<NZ, false>), dim3(grid), dim3(QSMC_BLOCK), 0, s, e0, e1, 0, x, ldx, n, w_in, \
                                  w_out, prev_norm, ea, outcome, ro);                                                   \
        else                                                                                                            \
            hipExtLaunchKernelGGL((k_update_tomo<NZ, true>), dim3(grid), dim3(QSMC_BLOCK), 0, s, e0, e1, 0, x, ldx, n, w_in,  \
                                  w_out, prev_norm, ea, outcome, ro);                                                   \
        break;
                switch (ea.nnz) { LT(1) LT(2) LT(3) LT(4) }
#undef LT
            } else {
                launch_update<QSMC_MODEL_TOMOGRAPHY>(h, vec2, grid, s, x, ldx, n, w_in, w_out, prev_norm, ea, outcome, ro);
            }
            break;
        }
#undef LAUNCH_U
    }
    HIP_TRY(h, hipGetLastError());
    rc = launch_reduce(h, ns, grid, ro, s);
    if (rc) return rc;
    if (ro.prefix_gate) {
        rc = resample_prefix(h, w_out, n, 0.0, h->spec.n_out, h->spec.seed, h->spec.epoch, s, true);
        if (rc) return rc;
    }
    return collect_stats(h, ns, stats_host, moments_host, n_mom, s);
}

int qsmc_lw_arm_prefix(qsmc_handle_t h, int32_t enabled, double ess_below, int64_t n_out, uint64_t seed, uint64_t epoch) {
    if (!h || (enabled && (n_out <= 0 || !(ess_below == ess_below)))) return QSMC_ERR_INVALID;
    h->spec.enabled = enabled;
    h->spec.thresh = ess_below;
    h->spec.n_out = n_out;
    h->spec.seed = seed;
    h->spec.epoch = epoch;
    return QSMC_OK;
}

int qsmc_lw_prefix_stats(qsmc_handle_t h, int64_t *n_queued, int64_t *n_adopted) {
    if (!h || !n_queued || !n_adopted) return QSMC_ERR_INVALID;
    *n_queued = h->spec.n_queued;
    *n_adopted = h->spec.n_adopted;
    return QSMC_OK;
}

int qsmc_update_multi(qsmc_handle_t h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                      const double *w_in, double *w_out, double prev_norm, const qsmc_expparam_t *exps,
                      const int64_t *outcomes, int32_t k, qsmc_update_stats_t *stats_host, double *moments_host,
                      qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->rsq.valid = 0; h->canon_next.kind = 0; h->expect_next = 0.0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !x || !w_out || !exps || !outcomes || !stats_host || n <= 0 || k < 1 || k > MULTI_KMAX)
        return QSMC_ERR_INVALID;
    int rc = check_model(model);
    if (rc) return rc;
    const int d = model->d;
    const bool wide = d > QSMC_MAX_D;
    // wide clouds (16 < d <= 64): windows of SPARSE measurement vectors only -- k_update_multi_tomo reads rows by index and
    // does not care how many there are; anything else: QSMC_ERR_UNSUPPORTED, the caller loops qsmc_update_fused
    TomoWideArgs wa[MULTI_KMAX];
    if (wide) {
        if (k < 2 || !(aligned16(x) && (!w_in || aligned16(w_in)) && aligned16(w_out) && (ldx % 2 == 0)))
            return QSMC_ERR_UNSUPPORTED;
        for (int j = 0; j < k; ++j) {
            rc = make_wide_args(model, &exps[j], &wa[j]);
            if (rc) return rc;
            if (wa[j].nnz < 1 || wa[j].nnz > MULTI_TOMO_NZ || wa[j].lik_pow != 0.0) return QSMC_ERR_UNSUPPORTED;
        }
    }
    const int dmom = d <= 4 ? d : 0;
    if (moments_host && !dmom) return QSMC_ERR_UNSUPPORTED;
    const int n_mom = dmom + dmom * (dmom + 1) / 2;
    const int ns = 3 * MULTI_KMAX + n_mom;
    hipStream_t s = (hipStream_t)stream;
    const int per_block = QSMC_BLOCK * MULTI_PER_THREAD;
    const int grid = grid_for(n, per_block);
    rc = ensure_partials(h, (size_t)grid * (ns + 1));
    if (rc) return rc;
    MultiArgs ma;
    memset(&ma, 0, sizeof(ma));
    ma.k = k;
    for (int j = 0; j < k && !wide; ++j) {
        make_exp_args(model, &exps[j], outcomes[j], &ma.e[j]);
        ma.outcome[j] = outcomes[j];
        if (outcomes[j] != 0) ma.outcome_mask |= 1u << j;   // two_outcome as fma(lb, pr0, la)
        const double ht = 0.5 * fabs(ma.e[j].t), wa = fabs(ma.e[j].w_);
        ma.half_tmax = (ht > ma.half_tmax || ht != ht) ? ht : ma.half_tmax;      // (a NaN time: every particle takes the general path)
        ma.wabs_max = (wa > ma.wabs_max || wa != wa) ? wa : ma.wabs_max;
    }
    if (g_multi_generic) ma.half_tmax = -1.0;                      // (test hook, qsmc_test_hook: every tile through the general path)
    ReduceOut ro = make_reduce(h, true, nullptr);
    // per-tile sums of the window's final weights + their chunk prefix, as qsmc_update_fused leaves them: a resample
    // after the window (qsmc_lw_use_update_sums with this call's token) does not read the weights again
    if (ensure_tile_sums(h, (size_t)((n + BUCKET_CHUNK - 1) / BUCKET_CHUNK) * (BUCKET_CHUNK / per_block) *
                                                QSMC_WAVES_PER_BLOCK) == QSMC_OK) {
        ro.tile_sums = h->tile_sums;
        h->ts.w = w_out;
        h->ts.n = n;
        h->ts.tile = per_block;
    }
    ++h->ts.gen;
    h->ts.armed = 0;
    h->spec.launched = 0;
    rc = setup_tile_prefix(h, ro, n, per_block, ns);
    if (rc) return rc;
    hipEvent_t me0 = nullptr, me1 = nullptr;
    prof_events(h, QSMC_PROF_UPDATE_MULTI, &me0, &me1);
    // tomography with sparse measurement vectors (every datum of the window touches at most four rows -- a Pauli
    // measurement two): the window reads those rows only (k_update_multi_tomo)
    bool sparse_tomo = false;
    int nz_max = 0;
    MultiTomoArgs mt;
    memset(&mt, 0, sizeof(mt));
    if (wide) {
        sparse_tomo = true;
        for (int j = 0; j < k; ++j) {
            for (int q = 0; q < wa[j].nnz; ++q) {
                mt.idx[j][q] = wa[j].idx[q];
                mt.mv[j][q] = wa[j].val[q];
            }
            mt.outcome[j] = outcomes[j];
            nz_max = wa[j].nnz > nz_max ? wa[j].nnz : nz_max;
        }
    } else if (model->kind == QSMC_MODEL_TOMOGRAPHY && k >= 2) {
        const bool dense_env = g_tomo_dense;                      // (test hook, as for the single datum)
        sparse_tomo = !dense_env && aligned16(x) && (!w_in || aligned16(w_in)) && aligned16(w_out) && (ldx % 2 == 0) &&
                      ma.e[0].lik_pow == 0.0;
        for (int j = 0; j < k && sparse_tomo; ++j) {
            if (ma.e[j].nnz < 1 || ma.e[j].nnz > MULTI_TOMO_NZ) sparse_tomo = false;
            nz_max = ma.e[j].nnz > nz_max ? ma.e[j].nnz : nz_max;
        }
        if (sparse_tomo) {
            for (int j = 0; j < k; ++j) {
                for (int q = 0; q < ma.e[j].nnz; ++q) {
                    mt.idx[j][q] = ma.e[j].nz_idx[q];
                    mt.mv[j][q] = ma.e[j].meas[ma.e[j].nz_idx[q]];
                }
                mt.outcome[j] = outcomes[j];                    // (padding: row 0, coefficient +0 -- the memset)
            }
        }
    }
    if (sparse_tomo) {
#define LMT(KK)                                                                                                       \
    case KK:                                                                                                          \
        if (nz_max <= 2)                                                                                              \
            hipExtLaunchKernelGGL((k_update_multi_tomo<KK, 2>), dim3(grid), dim3(QSMC_BLOCK), 0, s, me0, me1, 0, x, ldx, n, \
                                  w_in, w_out, prev_norm, mt, ro);                                                    \
        else                                                                                                          \
            hipExtLaunchKernelGGL((k_update_multi_tomo<KK, MULTI_TOMO_NZ>), dim3(grid), dim3(QSMC_BLOCK), 0, s, me0, me1, 0, \
                                  x, ldx, n, w_in, w_out, prev_norm, mt, ro);                                         \
        break;
            switch (k) { LMT(2) LMT(3) LMT(4) LMT(5) LMT(6) LMT(7) LMT(8) }
#undef LMT
    }
    if (!sparse_tomo)
    switch (model->kind) {
#define LAUNCH_MU(K)                                                                                   \
    case K:                                                                                            \
        if (ma.e[0].lik_pow != 0.0)                                                                    \
            hipExtLaunchKernelGGL((k_update_multi<K, true>), dim3(grid), dim3(QSMC_BLOCK), 0, s, me0, me1, 0, x, ldx, n, \
                                  w_in, w_out, prev_norm, ma, ro);                                     \
        else                                                                                           \
            hipExtLaunchKernelGGL((k_update_multi<K, false>), dim3(grid), dim3(QSMC_BLOCK), 0, s, me0, me1, 0, x, ldx, n, \
                                  w_in, w_out, prev_norm, ma, ro);                                     \
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
    return qsmc_hypothetical_sums_multi(h, model, x, ldx, n, w, norm, exp, 1, outcomes, &n_o, shift,
                                        QSMC_HYP_LOG | QSMC_HYP_MOMENTS, out_host, stream);
}

int qsmc_hypothetical_sums_begin(qsmc_handle_t h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                                 const double *w, double norm, const qsmc_expparam_t *exps, int32_t n_e,
                                 const int64_t *outcomes, const int32_t *n_o, const double *shift, int32_t what,
                                 double *out_host, qsmc_stream_t stream) {
    if (!h || !x || !exps || !outcomes || !n_o || !out_host || n <= 0 || n_e < 1) return QSMC_ERR_INVALID;
    if (what & ~(QSMC_HYP_LOG | QSMC_HYP_MOMENTS)) return QSMC_ERR_INVALID;
    static_assert(QSMC_HYP_LOG == HYP_WHAT_LOG && QSMC_HYP_MOMENTS == HYP_WHAT_MOM, "the header's bits are the kernels'");
    for (int e = 0; e < n_e; ++e) if (n_o[e] < 1) return QSMC_ERR_INVALID;
    int rc = check_model(model);
    if (rc) return rc;
    if (model->d > QSMC_MAX_D) return QSMC_ERR_UNSUPPORTED;   // (wide clouds: design through qsmc_likelihood)
    if (!h->hypq) {
        h->hypq = new (std::nothrow) Chain2Queue();
        if (!h->hypq) return QSMC_ERR_ALLOC;
    }
    hipStream_t s = (hipStream_t)stream;
    // a row: [N, sum w L ln L] and, where the kernels carry them (Model<KIND>::D <= 4 -- a compile-time dimension: tomography's
    // is its maximum, so a ONE-qubit tomography model, d = 4, has two columns like every other tomography model), the moments
    const int per = 2 + 2 * ((model->kind != QSMC_MODEL_TOMOGRAPHY && model->d <= 4) ? model->d : 0);
    Chain2Queue &q = *h->hypq;
    size_t done = 0;
    for (int e = 0; e < n_e && rc == QSMC_OK; ++e) {
        const int64_t *oc = outcomes + done;
        double *oh = out_host + done * per;
        switch (model->kind) {
#define HD(K) case K: rc = hyp_dispatch<K>(h, q, model, x, ldx, n, w, norm, exps + e, oc, n_o[e], shift, oh, s, what); break;
            HD(QSMC_MODEL_PRECESSION)
            HD(QSMC_MODEL_BINOMIAL_PRECESSION)
            HD(QSMC_MODEL_RB)
            HD(QSMC_MODEL_RB_INTERLEAVED)
            HD(QSMC_MODEL_BINOMIAL_RB)
            HD(QSMC_MODEL_BINOMIAL_RB_INTERLEAVED)
            HD(QSMC_MODEL_UNKNOWN_T2)
            HD(QSMC_MODEL_TOMOGRAPHY)
#undef HD
            default: rc = QSMC_ERR_INVALID;
        }
        done += (size_t)n_o[e];
    }
    if (rc) {                                  // leave nothing in flight that writes the pinned block behind the caller's back
        (void)hipStreamSynchronize(s);
        q.count = 0;
        q.off = 0;
    }
    return rc;
}

int qsmc_hypothetical_sums_collect(qsmc_handle_t h, qsmc_stream_t stream) {
    if (!h) return QSMC_ERR_INVALID;
    if (!h->hypq) return QSMC_OK;
    return chain2_flush(h, *h->hypq, (hipStream_t)stream);
}

int qsmc_hypothetical_sums_multi(qsmc_handle_t h, const qsmc_model_t *model, const double *x, int64_t ldx, int64_t n,
                                 const double *w, double norm, const qsmc_expparam_t *exps, int32_t n_e,
                                 const int64_t *outcomes, const int32_t *n_o, const double *shift, int32_t what,
                                 double *out_host, qsmc_stream_t stream) {
    const int rc = qsmc_hypothetical_sums_begin(h, model, x, ldx, n, w, norm, exps, n_e, outcomes, n_o, shift, what, out_host,
                                                stream);
    if (rc) return rc;
    return qsmc_hypothetical_sums_collect(h, stream);
}

int qsmc_update_from_likelihood(qsmc_handle_t h, const double *L, int64_t n, const double *w_in,
                                double *w_out, double prev_norm, double *stats_dev,
                                qsmc_update_stats_t *stats_host, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->rsq.valid = 0; h->canon_next.kind = 0; h->expect_next = 0.0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !L || !w_in || !w_out || n <= 0) return QSMC_ERR_INVALID;
    return weights_pass<0>(h, L, n, w_in, w_out, prev_norm, stats_dev, stats_host, (hipStream_t)stream);
}

int qsmc_clip_weights(qsmc_handle_t h, double *w, int64_t n, double norm, double *stats_dev,
                      qsmc_update_stats_t *stats_host, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->rsq.valid = 0; h->canon_next.kind = 0; h->expect_next = 0.0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
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

int qsmc_kde_cross_entropy(qsmc_handle_t h, const double *x, int64_t ldx, int64_t n, const double *w, double norm_p,
                           const double *y, int64_t ldy, int64_t m, const double *v, double norm_q, int32_t d,
                           const double *scale_host, double *out_host, qsmc_stream_t stream) {
    if (!h || !x || !y || !scale_host || !out_host || n <= 0 || m <= 0 || d < 1 || d > QSMC_MAX_D ||
        !(norm_p > 0.0) || !(norm_q > 0.0))
        return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n, QSMC_BLOCK);
    int rc = ensure_partials(h, (size_t)grid * 4);
    if (rc) return rc;
    KdeScale sc;
    for (int q = 0; q < QSMC_MAX_D; ++q) sc.s[q] = q < d ? scale_host[q] : 0.0;
    const ReduceOut ro = make_reduce(h, true, nullptr);
    hipLaunchKernelGGL(k_kde_cross, dim3(grid), dim3(QSMC_BLOCK), 0, s, x, ldx, n, w, 1.0 / norm_p, y, ldy, m, v,
                       1.0 / norm_q, (int)d, sc, ro);
    HIP_TRY(h, hipGetLastError());
    rc = launch_reduce(h, 3, grid, ro, s);
    if (rc) return rc;
    qsmc_update_stats_t st;
    rc = collect_stats(h, 3, &st, nullptr, 0, s);
    if (rc) return rc;
    *out_host = st.sum;
    return QSMC_OK;
}

int qsmc_normalize_weights(qsmc_handle_t h, const double *w_in, double *w_out, int64_t n, double norm,
                           qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->rsq.valid = 0; h->canon_next.kind = 0; h->expect_next = 0.0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !w_in || !w_out || n < 0) return QSMC_ERR_INVALID;
    if (n == 0) return QSMC_OK;
    return weights_pass<2>(h, nullptr, n, w_in, w_out, norm, nullptr, nullptr, (hipStream_t)stream);
}

int qsmc_fill(qsmc_handle_t h, double *w, int64_t n, double value, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->rsq.valid = 0; h->canon_next.kind = 0; h->expect_next = 0.0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !w || n < 0) return QSMC_ERR_INVALID;
    if (n == 0) return QSMC_OK;
    hipLaunchKernelGGL(k_fill, dim3(grid_for(n, QSMC_BLOCK * 4)), dim3(QSMC_BLOCK), 0, (hipStream_t)stream, w,
                       n, value);
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

// 16 < d <= 64: k_moments_wide<NB> (upper block triangle on the matrix cores), partial rows summed by k_sum_partials,
// packed on the host into qsmc_moments' layout
static int moments_wide(qsmc_ctx *h, const double *x, int64_t ldx, int64_t n, int d, const double *w, double norm,
                        double *out_dev, double *out_host, hipStream_t s) {
    const int nb = (d + 15) / 16, np = wide_pairs(nb), KW = wide_mom_k(nb), K = 1 + d + d * (d + 1) / 2;
    int gridm = grid_for(n, QSMC_WAVES_PER_BLOCK * 16 * 4);
    gridm = gridm < 768 ? gridm : 768;
    int rc = ensure_partials(h, (size_t)gridm * KW);
    if (rc) return rc;
    rc = ensure_scratch(h, 256 + (size_t)KW);
    if (rc) return rc;
    hipEvent_t m0 = nullptr, m1 = nullptr;
    prof_events(h, QSMC_PROF_MOMENTS, &m0, &m1);
    const int nt_m = ((double)n * 8.0 * (double)(d + 1) > 3.0e8) ? 1 : 0;
    switch (nb) {
        case 2: hipExtLaunchKernelGGL((k_moments_wide<2>), dim3(gridm), dim3(QSMC_BLOCK), 0, s, m0, m1, 0, x, ldx, n, d, w, norm, h->partials, nt_m); break;
        case 3: hipExtLaunchKernelGGL((k_moments_wide<3>), dim3(gridm), dim3(QSMC_BLOCK), 0, s, m0, m1, 0, x, ldx, n, d, w, norm, h->partials, nt_m); break;
        default: hipExtLaunchKernelGGL((k_moments_wide<4>), dim3(gridm), dim3(QSMC_BLOCK), 0, s, m0, m1, 0, x, ldx, n, d, w, norm, h->partials, nt_m); break;
    }
    double *full = h->scratch + 256;
    hipLaunchKernelGGL(k_sum_partials, dim3((KW + QSMC_WAVES_PER_BLOCK - 1) / QSMC_WAVES_PER_BLOCK), dim3(QSMC_BLOCK), 0, s,
                       h->partials, gridm, KW, full);
    HIP_TRY(h, hipGetLastError());
    double *hostfull = static_cast<double *>(malloc(((size_t)KW + (size_t)K) * sizeof(double)));
    if (!hostfull) return QSMC_ERR_ALLOC;
    rc = read_back(h, full, hostfull, (size_t)KW, s);
    if (rc) { free(hostfull); return rc; }
    double *packed = hostfull + KW;
    packed[0] = hostfull[np * 256 + 16 * nb];
    int k = 1 + d;
    for (int m = 0; m < d; ++m) {
        packed[1 + m] = hostfull[np * 256 + m];
        const int bi = m >> 4;
        for (int q = m; q < d; ++q) {
            const int bj = q >> 4;
            const int pair = bi * nb - bi * (bi - 1) / 2 + (bj - bi);          // (bi, bj), bi <= bj, row-major over the upper triangle
            packed[k++] = hostfull[pair * 256 + (m & 15) * 16 + (q & 15)];
        }
    }
    if (out_host) memcpy(out_host, packed, (size_t)K * sizeof(double));
    if (out_dev) {
        hipError_t e = hipMemcpyAsync(out_dev, packed, (size_t)K * sizeof(double), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { free(hostfull); HIP_TRY(h, e); }
    }
    free(hostfull);
    return QSMC_OK;
}

int qsmc_moments(qsmc_handle_t h, const double *x, int64_t ldx, int64_t n, int32_t d, const double *w,
                 double norm, double *out_dev, double *out_host, qsmc_stream_t stream) {
    if (!h || !x || !w || n <= 0 || d < 1 || d > QSMC_MAX_D_WIDE) return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int K = 1 + d + d * (d + 1) / 2;
    if (d > QSMC_MAX_D) return moments_wide(h, x, ldx, n, d, w, norm, out_dev, out_host, s);
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
    hipEvent_t m0 = nullptr, m1 = nullptr;
    prof_events(h, QSMC_PROF_MOMENTS, &m0, &m1);
    hipExtLaunchKernelGGL(k_moments_mfma, dim3(gridm), dim3(QSMC_BLOCK), 0, s, m0, m1, 0, x, ldx, n, d, w, norm, h->partials);
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

// a, mean (d) and S (d x d, row stride d) of a d > 16 resample: into the next pinned slot, then one H2D copy on `s`.  The
// slots are used in turn; an event behind each copy says when its slot may be rewritten (WIDE_RING calls later: the wait is
// a formality).  Null pointers leave that part of the slot zero.
static int upload_lw_wide(qsmc_ctx *h, int d, double a, const double *mean, const double *S, hipStream_t s) {
    const int si = h->lw_wide_next++ % WIDE_RING;
    LWWide *slot = h->lw_wide_host + si;
    if (!h->lw_wide_ev[si]) HIP_TRY(h, hipEventCreateWithFlags(&h->lw_wide_ev[si], hipEventDisableTiming));
    else HIP_TRY(h, hipEventSynchronize(h->lw_wide_ev[si]));        // (the copy that last read this slot has run)
    memset(slot, 0, sizeof(*slot));
    slot->a = a;
    if (mean) memcpy(slot->mean, mean, sizeof(double) * d);
    if (S) memcpy(slot->S, S, sizeof(double) * d * d);
    HIP_TRY(h, hipMemcpyAsync(h->lw_wide, slot, sizeof(LWWide), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipEventRecord(h->lw_wide_ev[si], s));
    return QSMC_OK;
}

int qsmc_lw_centres(qsmc_handle_t h, const double *x_in, int64_t ldx_in, int32_t d, const int64_t *js,
                    int64_t n_out, double a, const double *mean, double *mus, int64_t ld_mus,
                    qsmc_stream_t stream) {
    if (!h || !x_in || !js || !mean || !mus || d < 1 || d > QSMC_MAX_D_WIDE || n_out < 0) return QSMC_ERR_INVALID;
    if (n_out == 0) return QSMC_OK;
    if (d > QSMC_MAX_D) {
        int rc = upload_lw_wide(h, d, a, mean, nullptr, (hipStream_t)stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_centres_wide, dim3(grid_for(n_out, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, (hipStream_t)stream,
                           x_in, ldx_in, d, js, n_out, h->lw_wide, mus, ld_mus);
        HIP_TRY(h, hipGetLastError());
        return QSMC_OK;
    }
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
    if (model->d < 1 || model->d > QSMC_MAX_D_WIDE) return QSMC_ERR_INVALID;
    if (k == 0) return QSMC_OK;
    if (model->d > QSMC_MAX_D) {
        if (model->kind != QSMC_MODEL_TOMOGRAPHY) return QSMC_ERR_UNSUPPORTED;      // (no validity test above QSMC_MAX_D)
        int rc = upload_lw_wide(h, model->d, 0.0, nullptr, S, (hipStream_t)stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_perturb_wide, dim3(grid_for(k, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, (hipStream_t)stream,
                           model->d, mus, ld_mus, idxs, k, centre_by_idx, h->lw_wide, z, ldz, x_out, ldx_out, valid_out);
        HIP_TRY(h, hipGetLastError());
        return QSMC_OK;
    }
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
// (d = 16 only, in a buffer of their own -- the prefix is queued before d is known and must not be reallocated under the
//  kernels that use it: anc[n_out] u32 | canon count[4] + list[n_out] u32)
struct BucketPlan {
    bool bucketed;
    int chunks, max_items, cap;
    unsigned int *hist, *counts, *retry_list, *anc, *clist;
    long long *slot_off;
    int *item_off, *item_chunk;
};

// Outputs per work item.  A work item re-scans its chunk's weights, so big items amortise that -- but the grid must fill
// the chip: 768 workgroups are resident at once (256 CUs x 3), and a cloud of 1.25e6 particles (the per-GPU share of
// config 5) has only 306 chunks, i.e. 1.2 workgroups per CU at 8192 outputs each: the d = 16 sampler ran at a third
// of a wave's worth of latency hiding (383 us).  Largest power of two <= 8192 that still leaves >= 1024 items.
static int bucket_cap(int64_t n_out) {
    int cap = BUCKET_CAP;
    while (cap > 512 && n_out / cap < 1024) cap >>= 1;
    return cap;
}

static bool use_buckets(int64_t chunks64, int64_t n_out) {
    return chunks64 <= BUCKET_MAX_CHUNKS && n_out >= 4 * BUCKET_CHUNK && n_out < (1ll << 32);
}

static int ensure_anc16(qsmc_ctx *h, size_t bytes) {
    if (h->anc16_cap >= bytes) return QSMC_OK;
    if (h->anc16) HIP_TRY(h, hipFree(h->anc16));
    h->anc16 = nullptr;
    h->anc16_cap = 0;
    HIP_TRY(h, hipMalloc(&h->anc16, bytes));
    h->anc16_cap = bytes;
    return QSMC_OK;
}

static int bucket_plan_layout(qsmc_ctx *h, int64_t chunks64, int64_t n_out, BucketPlan *bp, bool split16 = false) {
    bp->bucketed = use_buckets(chunks64, n_out);
    bp->anc = bp->clist = nullptr;
    if (!bp->bucketed) return QSMC_OK;
    const int chunks = (int)chunks64;
    const size_t hist_b = (size_t)BUCKET_COUNT_BLOCKS * chunks * sizeof(unsigned int);
    const size_t counts_b = ((size_t)chunks * sizeof(unsigned int) + 15) & ~(size_t)15;
    const size_t slot_b = ((size_t)(chunks + 1) * sizeof(long long) + 15) & ~(size_t)15;
    const size_t item_b = ((size_t)(chunks + 1) * sizeof(int) + 15) & ~(size_t)15;
    const int cap = bucket_cap(n_out);
    const int max_items = chunks + (int)(n_out / cap) + 1;
    const size_t map_b = ((size_t)max_items * sizeof(int) + 15) & ~(size_t)15;
    const size_t retry_b = ((size_t)n_out * sizeof(unsigned int) + 15) & ~(size_t)15;
    int rc = ensure_iscratch(h, hist_b + counts_b + slot_b + item_b + map_b + retry_b);
    if (rc) return rc;
    if (split16) rc = ensure_anc16(h, 3 * retry_b + 64);        // ancestors, canonicalize's list (+ 4 count words), its second list
    if (rc) return rc;
    unsigned char *basep = reinterpret_cast<unsigned char *>(h->iscratch);
    bp->chunks = chunks;
    bp->max_items = max_items;
    bp->cap = cap;
    bp->hist = reinterpret_cast<unsigned int *>(basep);
    bp->counts = reinterpret_cast<unsigned int *>(basep + hist_b);
    bp->slot_off = reinterpret_cast<long long *>(basep + hist_b + counts_b);
    bp->item_off = reinterpret_cast<int *>(basep + hist_b + counts_b + slot_b);
    bp->item_chunk = reinterpret_cast<int *>(basep + hist_b + counts_b + slot_b + item_b);
    bp->retry_list = reinterpret_cast<unsigned int *>(basep + hist_b + counts_b + slot_b + item_b + map_b);
    if (split16) {
        bp->anc = h->anc16;
        bp->clist = reinterpret_cast<unsigned int *>(reinterpret_cast<unsigned char *>(h->anc16) + retry_b);
    }
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
// speculative = true: called by qsmc_update_fused itself right behind the reducing kernel (qsmc_lw_arm_prefix): the
// count kernel is gated on the device-side resample test and takes the normaliser from device memory; only the
// one-launch form (tile sums -> edges -> counts -> plan) is ever queued this way.
static int resample_prefix(qsmc_ctx *h, const double *w, int64_t n_in, double norm, int64_t n_out, uint64_t seed,
                           uint64_t epoch, hipStream_t s, bool speculative) {
    if (!speculative && h->spec.launched) {
        // the update that wrote these weights queued this very prefix behind itself and its gate opened: done already
        // (everything the host would pass now is compared with what the device used; any difference -> redo it here)
        h->spec.launched = 0;
        if (h->spec.gen == h->ts.gen && h->ts.armed == h->ts.gen && w && w == h->spec.w && h->ts.w == w &&
            n_in == h->spec.n_in &&
            n_out == h->spec.n_out && seed == h->spec.seed && epoch == h->spec.epoch &&
            h->mapped[REDUCE_OUT_MAX - 3] == 1.0 && h->mapped[REDUCE_OUT_MAX - 4] == norm) {
            h->ts.armed = 0;
            ++h->spec.n_adopted;
            return QSMC_OK;
        }
    }
    uint32_t k0, k1, ep;
    philox_keys(seed, epoch, &k0, &k1, &ep);
    const int64_t chunks64 = (n_in + BUCKET_CHUNK - 1) / BUCKET_CHUNK;
    int rc = ensure_rs_offsets(h, (size_t)chunks64 + 1);
    if (rc) return rc;
    double *offsets = h->rs_offsets;
    const double inv_norm = speculative ? 0.0 : 1.0 / norm;
    // chunk sums: from the tile sums of the update that produced these very weights, if the caller vouches for
    // that (qsmc_lw_use_update_sums) and nothing has touched them since; else one pass over the weights
    TileSrc ts{nullptr, 0, 0, 0.0};
    if ((speculative || (h->ts.armed && h->ts.armed == h->ts.gen)) && w && h->ts.w == w && h->ts.n == n_in)
        ts = TileSrc{h->tile_sums, BUCKET_CHUNK / h->ts.tile * QSMC_WAVES_PER_BLOCK,
                     (n_in + h->ts.tile - 1) / h->ts.tile * QSMC_WAVES_PER_BLOCK, inv_norm};
    if (!speculative) h->ts.armed = 0;
    BucketPlan bp;
    rc = bucket_plan_layout(h, chunks64, n_out, &bp);
    if (rc) return rc;
    // (k_bucket_counts meets at grid barriers: its 16 workgroups need a CU each -- any real part has them; under a CU
    //  mask that leaves fewer, the counts come from the one-uniform-per-output histogram instead: same law, no barrier)
    const bool count_by_draws = h->cu_count < BUCKET_COUNTS_BLOCKS;
    // with tile sums the bucketed count kernel forms the offsets itself; otherwise: chunk sums, then the scan
    const bool scan_in_counts = ts.tiles && bp.bucketed && !count_by_draws;
    if (speculative && !scan_in_counts) return QSMC_OK;         // nothing queued (h->spec.launched stays 0)
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
            hipLaunchKernelGGL(k_bucket_plan, dim3(1), dim3(1024), 0, s, bp.counts, chunks, bp.cap, bp.slot_off, bp.item_off,
                               bp.item_chunk);
        } else {
            const double kappa = g_poisson_margin;                            // (5; test hook: 0 makes the removal branch common)
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
            hipEvent_t q0 = nullptr, q1 = nullptr;
            const int ring_before = h->prof_n;
            prof_events(h, QSMC_PROF_COUNTS, &q0, &q1);
            hipExtLaunchKernelGGL(k_bucket_counts, dim3(BUCKET_COUNTS_BLOCKS), dim3(BUCKET_COUNTS_THREADS), lds, s, q0, q1, 0, offsets,
                               scan_in_counts ? ts : TileSrc{nullptr, 0, 0, 0.0},
                               (scan_in_counts && h->tile_prefix_gen == h->ts.gen) ? h->tile_prefix : (const double *)nullptr,
                               reinterpret_cast<unsigned long long *>(h->counter), chunks, n_out, lambda, k0, k1, ep,
                               bp.counts, extra, bp.slot_off, bp.item_off, bp.item_chunk, h->gbar, h->gbar_base, bp.cap,
                               speculative ? h->spec.gate : (const int *)nullptr,
                               speculative ? h->red_out : (const double *)nullptr);
            h->gbar_base += 2ull * BUCKET_COUNTS_BLOCKS;
            if (speculative) {
                h->spec.launched = 1;
                ++h->spec.n_queued;
                h->spec.gen = h->ts.gen;
                h->spec.w = w;
                h->spec.n_in = n_in;
                h->spec.prof_slot = (q0 && h->prof_n != ring_before) ? ring_before % QSMC_PROF_CAP : -1;
            }
        }
    }
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

static unsigned long long splitmix64(unsigned long long &x) {
    unsigned long long z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Device buffers of the proposal bank for a resample of n_out particles over max_items work items, expecting `lambda`
// spares: entries (2 lambda + 65536 of them: the Poisson total does not get there), the per-item arrays and prefixes,
// the counters, two round lists with their block counts / prefixes, the leftover list.
constexpr int BANK_STRIDE_MAX = 8;
static int bank_layout(qsmc_ctx *h, double lambda, int max_items, int64_t n_out, uint64_t seed, uint64_t epoch, BankOut *bo,
                       BankIn *bi) {
    const long long capacity = (long long)(2.0 * lambda) + 65536;
    if (h->bank.capacity < capacity) {
        if (h->bank.entries) HIP_TRY(h, hipFree(h->bank.entries));
        h->bank.entries = nullptr;
        h->bank.capacity = 0;
        HIP_TRY(h, hipMalloc(&h->bank.entries, (size_t)capacity * BANK_STRIDE_MAX * sizeof(double)));
        h->bank.capacity = capacity;
    }
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t items_i = up((size_t)max_items * sizeof(int)), items_l = up(((size_t)max_items + 1) * sizeof(long long));
    const size_t nvb = (size_t)(n_out / BANK_VB) + 2;
    const size_t list_b = up((2 * (size_t)n_out + 8192) * sizeof(unsigned int));   // (k_bank_tail splits a list buffer in two)
    const size_t need = 256 /* top */ + up(64 * sizeof(long long)) + 2 * items_i + 3 * items_l + 2 * list_b +
                        4 * up(nvb * sizeof(int)) + 2 * up((nvb + 1) * sizeof(long long)) + list_b;
    if (h->bank.aux_cap < need) {
        if (h->bank.aux) HIP_TRY(h, hipFree(h->bank.aux));
        h->bank.aux = nullptr;
        h->bank.aux_cap = 0;
        HIP_TRY(h, hipMalloc(&h->bank.aux, need));
        HIP_TRY(h, hipMemset(h->bank.aux, 0, need));          // (the counters start at zero; the kernels leave them so)
        h->bank.aux_cap = need;
    }
    unsigned char *p = h->bank.aux;
    auto take = [&](size_t b) { unsigned char *r = p; p += b; return r; };
    bo->lambda = lambda;
    bo->stride = bi->stride = BANK_STRIDE_MAX;
    bo->entries = h->bank.entries;
    bo->capacity = h->bank.capacity;
    (void)take(256);
    bi->ctr = reinterpret_cast<long long *>(take(up(64 * sizeof(long long))));
    bo->e_cnt = reinterpret_cast<int *>(take(items_i));
    bo->f_cnt = reinterpret_cast<int *>(take(items_i));
    bo->f_base = reinterpret_cast<long long *>(take(items_l));
    bi->e_off = reinterpret_cast<long long *>(take(items_l));
    bi->f_off = reinterpret_cast<long long *>(take(items_l));
    for (int k = 0; k < 2; ++k) bi->tmp[k] = reinterpret_cast<unsigned int *>(take(list_b));
    for (int k = 0; k < 2; ++k) bi->bcount[k] = reinterpret_cast<int *>(take(up(nvb * sizeof(int))));
    for (int k = 0; k < 2; ++k) bi->vb_first[k] = reinterpret_cast<int *>(take(up(nvb * sizeof(int))));
    for (int k = 0; k < 2; ++k) bi->boff[k] = reinterpret_cast<long long *>(take(up((nvb + 1) * sizeof(long long))));
    bi->leftover = reinterpret_cast<unsigned int *>(take(list_b));
    bi->entries = bo->entries;
    bi->e_cnt = bo->e_cnt;
    bi->f_cnt = bo->f_cnt;
    bo->e_base = bi->e_off;                          // an item's spares start at the prefix of the counts: spare g is entry g
    bi->f_base = bo->f_base;
    unsigned long long sm = seed ^ (epoch * 0xD1342543DE82EF95ull) ^ 0x62616E6Bull;       // "bank"
    for (int k = 0; k < 4; ++k) bi->key[k] = splitmix64(sm);
    return QSMC_OK;
}

// TomographyModel.canonicalize (smc.py:529) folded into a d = 16 resample: kind 0 = not asked for
struct CanonSpec {
    int kind, allow_sub;
    const double *basis;
};
constexpr int RS_STAGE_ANCESTORS = 1, RS_STAGE_KICK = 2, RS_STAGE_ALL = 3;

// stages: the split d = 16 sampler may be queued in two halves (qsmc_step: ancestors while the host forms S, kicks after);
// every other caller passes RS_STAGE_ALL.
// 16 < d <= 64 (tomography: no validity test, so no redraws and no failures).  Ancestors as for d = 16 -- the weight-only
// prefix (or the one already queued: qsmc_lw_resample_prepare / the speculative one) and k_bucket_anc16 where the bucketed
// sampler applies, else the global CDF and one search per slot --, then the kicks on the matrix cores (k_kick_wide).
static int resample_philox_wide(qsmc_ctx *h, const double *x_in, int64_t ldx_in, int64_t n_in, int d, const double *w,
                                double norm, double a, const double *mean, const double *S, int64_t n_out, uint64_t seed,
                                uint64_t epoch, double *x_out, const OutPlace &pl, int64_t *n_failed_host, hipStream_t s) {
    if (n_in >= (1ll << 32)) return QSMC_ERR_UNSUPPORTED;          // (ancestors travel as 32-bit indices)
    uint32_t k0, k1, ep;
    philox_keys(seed, epoch, &k0, &k1, &ep);
    const int64_t chunks64 = (n_in + BUCKET_CHUNK - 1) / BUCKET_CHUNK;
    const bool prepared = h->prep.valid && h->prep.w == w && h->prep.n_in == n_in && h->prep.n_out == n_out &&
                          h->prep.norm == norm && h->prep.seed == seed && h->prep.epoch == epoch && h->prep.stream == s;
    h->prep.valid = 0;
    int rc = QSMC_OK;
    if (!prepared) {
        rc = resample_prefix(h, w, n_in, norm, n_out, seed, epoch, s, false);
        if (rc) return rc;
    }
    rc = upload_lw_wide(h, d, a, mean, S, s);
    if (rc) return rc;
    const double inv_norm = 1.0 / norm;
    BucketPlan bp;
    rc = bucket_plan_layout(h, chunks64, n_out, &bp, true);
    if (rc) return rc;
    const unsigned int *anc = nullptr;
    if (bp.bucketed) {
        hipEvent_t a0 = nullptr, a1 = nullptr;
        prof_events(h, QSMC_PROF_ANCESTORS, &a0, &a1);
        SqrtJob sj;
        memset(&sj, 0, sizeof(sj));
        hipExtLaunchKernelGGL((k_bucket_anc16<512>), dim3(bp.max_items), dim3(512), 0, s, a0, a1, 0, n_in, w, inv_norm,
                              h->rs_offsets, bp.chunks, bp.slot_off, bp.item_off, bp.item_chunk, k0, k1, ep, bp.anc, bp.cap,
                              bp.clist, sj);
        anc = bp.anc;
    } else {
        rc = ensure_cdf(h, (size_t)n_in);
        if (rc) return rc;
        rc = ensure_anc16(h, (size_t)n_out * sizeof(unsigned int) + 64);
        if (rc) return rc;
        hipLaunchKernelGGL(k_chunk_scan, dim3((unsigned)chunks64), dim3(SCAN_THREADS), 0, s, w, n_in, inv_norm, h->rs_offsets,
                           h->cdf_scratch, (const unsigned long long *)nullptr);
        hipLaunchKernelGGL(k_anc_direct, dim3(grid_for(n_out, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, s, h->cdf_scratch, n_in,
                           n_out, k0, k1, ep, h->anc16);
        anc = h->anc16;
    }
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    prof_events(h, QSMC_PROF_SAMPLE, &pe0, &pe1);
    const unsigned kgrid = (unsigned)((n_out + KICKW_PER_BLOCK - 1) / KICKW_PER_BLOCK);
    const int nb = (d + 15) / 16;
#define LAUNCH_KW(NB_, DIRECT_)                                                                                       \
    hipExtLaunchKernelGGL((k_kick_wide<NB_, DIRECT_>), dim3(kgrid), dim3(KICKW_BT), 0, s, pe0, pe1, 0, x_in, ldx_in, anc, \
                          n_out, d, h->lw_wide, k0, k1, ep, x_out, pl)
    if (bp.bucketed) {
        if (nb == 2) LAUNCH_KW(2, false); else if (nb == 3) LAUNCH_KW(3, false); else LAUNCH_KW(4, false);
    } else {
        if (nb == 2) LAUNCH_KW(2, true); else if (nb == 3) LAUNCH_KW(3, true); else LAUNCH_KW(4, true);
    }
#undef LAUNCH_KW
    HIP_TRY(h, hipGetLastError());
    if (n_failed_host) {
        HIP_TRY(h, hipStreamSynchronize(s));
        *n_failed_host = 0;
    }
    return QSMC_OK;
}

static int resample_philox_impl(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect,
                                const double *x_in, int64_t ldx_in, int64_t n_in, int32_t d, const double *w,
                                double norm, double a, const double *mean, const double *S, int64_t n_out,
                                uint64_t seed, uint64_t epoch, int32_t maxiter, double *x_out, const OutPlace &pl,
                                int64_t *n_failed_host, qsmc_stream_t stream, CanonSpec canon = CanonSpec{0, 0, nullptr},
                                int stages = RS_STAGE_ALL, double expect_redraws = 0.0,
                                const SqrtJob *sqrt_job = nullptr, const LWDev *lw_dev = nullptr) {
    if (!h || !model || !x_in || !mean || !S || !x_out || n_in <= 0 || n_out <= 0) return QSMC_ERR_INVALID;
    if (d != model->d || d < 1 || d > QSMC_MAX_D_WIDE || maxiter < 1) return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (d > QSMC_MAX_D) {
        if (model->kind != QSMC_MODEL_TOMOGRAPHY || stages != RS_STAGE_ALL || canon.kind != 0 || sqrt_job || lw_dev)
            return QSMC_ERR_UNSUPPORTED;
        h->rsq.valid = 0;
        return resample_philox_wide(h, x_in, ldx_in, n_in, d, w, norm, a, mean, S, n_out, seed, epoch, x_out, pl,
                                    n_failed_host, s);
    }
    if (h->rsq.valid && stages == RS_STAGE_ALL) {
        // qsmc_step queued a resample when its n_ess test failed: if this is that very call -- every argument equal,
        // mean and S bit for bit -- the work is done (or under way on `stream`); anything else runs as if nothing had
        // been queued (the queued one wrote its own buffer and is overwritten or ignored)
        const auto &q = h->rsq;
        const bool same = q.model.kind == model->kind && q.model.d == model->d && q.model.min_freq == model->min_freq &&
                          q.postselect == postselect && q.d == d && q.maxiter == maxiter && q.x_in == x_in && q.w == w &&
                          q.ldx_in == ldx_in && q.n_in == n_in && q.n_out == n_out && q.norm == norm && q.a == a &&
                          q.seed == seed && q.epoch == epoch && q.x_out == x_out && q.stream == s && pl.n_dest == 0 &&
                          pl.ld_m == q.ldx_out && memcmp(q.mean, mean, sizeof(double) * d) == 0 &&
                          memcmp(q.S, S, sizeof(double) * d * d) == 0 && q.expect == expect_redraws &&
                          q.canon_kind == canon.kind &&
                          (canon.kind == 0 || (q.canon_allow_sub == canon.allow_sub && q.canon_basis == canon.basis));
        h->rsq.valid = 0;
        if (same) {
            ++h->rsq.n_adopted;
            if (n_failed_host) return read_counter(h, n_failed_host, s);
            return QSMC_OK;
        }
    }
    LWArgs lw;
    fill_lw(&lw, d, a, mean, S);
    uint32_t k0, k1, ep;
    philox_keys(seed, epoch, &k0, &k1, &ep);
    const int64_t chunks64 = (n_in + BUCKET_CHUNK - 1) / BUCKET_CHUNK;
    const bool split16 = d == 16 && model->kind == QSMC_MODEL_TOMOGRAPHY && use_buckets(chunks64, n_out);
    if (!split16 && (stages != RS_STAGE_ALL || canon.kind != 0)) return QSMC_ERR_UNSUPPORTED;
    if (canon.kind != 0 && pl.n_dest != 0) return QSMC_ERR_UNSUPPORTED;
    const bool prepared = h->prep.valid && h->prep.w == w && h->prep.n_in == n_in && h->prep.n_out == n_out &&
                          h->prep.norm == norm && h->prep.seed == seed && h->prep.epoch == epoch &&
                          h->prep.stream == s;
    h->prep.valid = 0;
    int rc = QSMC_OK;
    if (!prepared && (stages & RS_STAGE_ANCESTORS)) {
        rc = resample_prefix(h, w, n_in, norm, n_out, seed, epoch, s, false);
        if (rc) return rc;
    }
    double *offsets = h->rs_offsets;
    const double inv_norm = 1.0 / norm;
    unsigned long long *nf = reinterpret_cast<unsigned long long *>(h->counter);
    unsigned long long *retry_count = nf + 1;
    if (!split16) {
        rc = ensure_cdf(h, (size_t)n_in);
        if (rc) return rc;
    }
    BucketPlan bp;
    rc = bucket_plan_layout(h, chunks64, n_out, &bp, split16);      // no reallocation: the prefix sized it
    if (rc) return rc;
    if (!bp.bucketed) {
        // small or very large clouds: materialise the CDF and search it directly
        hipLaunchKernelGGL(k_chunk_scan, dim3((unsigned)chunks64), dim3(SCAN_THREADS), 0, s, w, n_in, inv_norm, offsets,
                           h->cdf_scratch, (const unsigned long long *)nullptr);
        hipLaunchKernelGGL(k_resample_philox, dim3(grid_for(n_out, QSMC_BLOCK)), dim3(QSMC_BLOCK), 0, s,
                           model->kind, d, model->min_freq, postselect, x_in, ldx_in, n_in, h->cdf_scratch, lw, n_out,
                           k0, k1, ep, maxiter, x_out, pl, nf);
    } else if (split16) {
        // 2-qubit tomography: ancestors (needs the weights and the plan only), then the kicks on the f64 matrix cores
        // with the first pass of canonicalize on the way out; tomography has no validity constraint: no redraws
        const int chunks = bp.chunks;
        if (stages & RS_STAGE_ANCESTORS) {
            hipEvent_t a0 = nullptr, a1 = nullptr;
            prof_events(h, QSMC_PROF_ANCESTORS, &a0, &a1);
            SqrtJob sj;
            memset(&sj, 0, sizeof(sj));
            if (sqrt_job) sj = *sqrt_job;
            hipExtLaunchKernelGGL((k_bucket_anc16<512>), dim3(bp.max_items + (sj.full ? 1 : 0)), dim3(512), 0, s, a0, a1, 0, n_in,
                                  w, inv_norm, offsets, chunks, bp.slot_off, bp.item_off, bp.item_chunk, k0, k1, ep, bp.anc,
                                  bp.cap, bp.clist /* [0]: the canonicalize list's length */, sj);
        }
        if (stages & RS_STAGE_KICK) {
            hipEvent_t pe0 = nullptr, pe1 = nullptr;
            prof_events(h, QSMC_PROF_SAMPLE, &pe0, &pe1);
            const int64_t n_ranges = (n_out + KICK16_PER_BLOCK - 1) / KICK16_PER_BLOCK;
            const unsigned kgrid = (unsigned)(((n_ranges + 7) / 8) * 8);
            unsigned int *ccount = bp.clist, *clist = bp.clist + 4;      // (ccount was cleared by k_bucket_anc16)
#define LAUNCH_K16(C)                                                                                                 \
    hipExtLaunchKernelGGL((k_bucket_kick16<C>), dim3(kgrid), dim3(KICK16_BT), 0, s, pe0, pe1, 0, x_in, ldx_in, bp.anc,  \
                          n_out, lw, k0, k1, ep, x_out, pl, canon.basis, canon.allow_sub, clist, ccount, lw_dev)
            if (canon.kind == 1) LAUNCH_K16(1);
            else if (canon.kind == 2) LAUNCH_K16(2);
            else LAUNCH_K16(0);
#undef LAUNCH_K16
            if (canon.kind) {
                hipEvent_t l0 = nullptr, l1 = nullptr;
                prof_events(h, QSMC_PROF_CANON_LIST, &l0, &l1);
                const int lgrid = grid_for(n_out, QSMC_BLOCK);
                unsigned int *clist2 = clist + ((size_t)n_out + 3) / 4 * 4 + 4;
                if (canon.kind == 1)
                    launch_canon_list4(TomoPauli2{}, lgrid, s, l0, l1, x_out, pl.ld_m, canon.allow_sub, clist, ccount, clist2);
                else
                    launch_canon_list4(TomoDense<4>{canon.basis}, lgrid, s, l0, l1, x_out, pl.ld_m, canon.allow_sub, clist,
                                       ccount, clist2);
            }
        }
    } else {
        const int chunks = bp.chunks;
        // 512-thread workgroups, CDF chunk + guide in LDS (40 KB -> 3 resident workgroups per CU, so one
        // workgroup's scan phase overlaps another's sampling loop); x is gathered from the chunk's
        // 32 KB global window (L2-resident).
        hipEvent_t pe0 = nullptr, pe1 = nullptr;
        prof_events(h, QSMC_PROF_SAMPLE, &pe0, &pe1);
#define LAUNCH_B(KERNEL, DD, BT)                                                                               \
    hipExtLaunchKernelGGL((KERNEL<DD, BT>), dim3(bp.max_items), dim3(BT), 0, s, pe0, pe1, 0, model->kind, d, \
                       model->min_freq, postselect, x_in, ldx_in, n_in, w, inv_norm, offsets,                       \
                       chunks, bp.slot_off, bp.item_off, bp.item_chunk, lw, k0, k1, ep,                             \
                       maxiter, x_out, pl, nf, bp.retry_list, retry_count, bp.cap)
#define LAUNCH_O(DD)                                                                                           \
    hipExtLaunchKernelGGL((k_bucket_sample_ordered<DD, 512>), dim3(bp.max_items), dim3(512), 0, s, pe0, pe1, 0, model->kind, d, \
                       model->min_freq, postselect, x_in, ldx_in, n_in, w, inv_norm, offsets,                       \
                       chunks, bp.slot_off, bp.item_off, bp.item_chunk, lw, k0, k1, ep,                             \
                       maxiter, x_out, pl, nf, bp.retry_list, retry_count, bp.cap, bo)
        // The proposal bank (kernels/resample.hpp): when the caller expects redraws (the count of this cloud's previous
        // resample), the ordered sampler also produces ~1.25 x that many spare proposals and the failed first tries are
        // served from them; the global-CDF redraw kernel stays behind it for whatever is left.
        BankOut bo;
        memset(&bo, 0, sizeof(bo));
        BankIn bi;
        memset(&bi, 0, sizeof(bi));
        bool banked = false;
        if (expect_redraws > 0.0 && postselect && maxiter > 1 && (d == 3 || d == 4)) {
            const double m = 1.25 * expect_redraws;
            const double lambda = m + 6.0 * sqrt(m) + 64.0;
            rc = bank_layout(h, lambda, bp.max_items, n_out, seed, epoch, &bo, &bi);
            if (rc) return rc;
            bo.stride = bi.stride = bank_stride(d);
            // (no memset command here: k_bank_scan clears the counters for the next resample -- a fill command between the
            //  plan and the sampler showed up as a 26 us bubble in the kernel trace)
            hipLaunchKernelGGL(k_bank_counts, dim3((bp.max_items + QSMC_BLOCK / POISSON_G - 1) / (QSMC_BLOCK / POISSON_G)),
                               dim3(QSMC_BLOCK), 0, s, offsets, chunks, bp.item_off, bp.item_chunk, bp.max_items, lambda, k0, k1,
                               ep, bo.e_cnt, bi.e_off, bi.ctr, bo.capacity);
            banked = true;
            ++h->bank.n_banked;
        }
        switch (d) {
            // d <= 2: the single-pass kernel; d >= 3: ancestors first, kicked in ascending order (coalesced gathers)
            case 1: LAUNCH_B(k_bucket_sample, 1, 512); break;
            case 2: LAUNCH_B(k_bucket_sample, 2, 512); break;
            case 3: LAUNCH_O(3); break;
            case 4: LAUNCH_O(4); break;
            default: LAUNCH_O(0); break;          // other d up to 16: runtime-d kernel
        }
#undef LAUNCH_B
#undef LAUNCH_O
        const unsigned int *redraw_list = bp.retry_list;
        const unsigned long long *redraw_count = retry_count;
        if (banked) {
            // four launches: prefixes, two device-wide rounds (each ends with its own prefix), the remaining rounds in one
            // workgroup.  (First cut: a launch per round and per prefix, 17 of them at ~5 us each -- as slow as the
            // global-CDF redraws they replace.)
            hipLaunchKernelGGL(k_bank_scan, dim3(1), dim3(1024), 0, s, bi, bp.item_off, chunks);
            double est = expect_redraws * 1.5 + 4096.0;
            for (int t = 1; t <= 2; ++t) {
                long long g = (long long)(est / BANK_VB) + 8;
                g = g > 8192 ? 8192 : g;
                hipLaunchKernelGGL((k_bank_round<4>), dim3((unsigned)g), dim3(BANK_VB), 0, s, bi, bp.item_off, chunks, t, d,
                                   bp.retry_list, x_out, pl);
                est *= 0.2;
            }
            hipLaunchKernelGGL((k_bank_tail<4>), dim3(1), dim3(1024), 0, s, bi, bp.item_off, chunks, 3, d, x_out, pl);
            redraw_list = bi.leftover;
            redraw_count = reinterpret_cast<const unsigned long long *>(bi.ctr + 2);
        }
        if (postselect && maxiter > 1) {
            // only if some particle asked for a global redraw do these two do any work
            // (more than two processes on one GPU -- bench.py's control-flow check -- must shrink the grid: all of
            //  them have to be resident together, 512 workgroup slots in total)
            static const int redraw_env = [] {
                const char *e = getenv("QSMC_REDRAW_BLOCKS");
                return e ? atoi(e) : REDRAW_BLOCKS;
            }();
            // never more than one workgroup per CU of THIS device (a partitioned part has fewer CUs than 256): the
            // grid must be resident as a whole, and half of the two-per-CU slots stay free for a second process
            int redraw_blocks = redraw_env < h->cu_count ? redraw_env : h->cu_count;
            redraw_blocks = redraw_blocks < 1 ? 1 : (redraw_blocks > REDRAW_BLOCKS ? REDRAW_BLOCKS : redraw_blocks);
            // Which form?  Nothing queued (most resamples of most models): ONE resident-grid launch that leaves at once.
            // Models whose postselection bites at every resample (RB: the cloud leans on A + B <= 1) spent ~300 us in
            // that kernel's scan phase -- 256 workgroups walking 3000 chunks in a dozen rounds behind a grid barrier --
            // so if the PREVIOUS resample on this handle queued redraws (its count came back with the last
            // host-visible reduction), the CDF is materialised by a full-grid gated k_chunk_scan first (40 us when
            // the gate is open, ~5 us when it is not) and the redraw kernel skips its scan and its barrier.  Same CDF,
            // same particles either way.
            const bool expect_cdf = !banked && h->mapped[REDUCE_OUT_MAX - 2] > 0.0;
            if (expect_cdf)
                hipLaunchKernelGGL(k_chunk_scan, dim3((unsigned)chunks64), dim3(SCAN_THREADS), 0, s, w, n_in, inv_norm,
                                   offsets, h->cdf_scratch, (const unsigned long long *)retry_count);
            // the chunk edges ride in LDS (first level of the redraw's ancestor search) while they fit 48 KB
            const size_t edges_lds = (size_t)(chunks64 + (chunks64 >> 5) + (chunks64 >> 10) + 4) * sizeof(double);      // (lds_skew)
            const int edges_in_lds = edges_lds <= 48 * 1024 ? 1 : 0;
            // (few queued outputs: chunk by chunk in LDS, no global CDF -- needs room for one chunk's CDF in the dynamic
            //  segment; QSMC_REDRAW_NO_SMALL keeps the global form for A/B)
            const bool no_small = g_redraw_no_small;                   // (test hook: the global form, for the same-bits test)
            const size_t small_bytes = (size_t)SCAN_CHUNK * sizeof(double);
            size_t dyn = edges_in_lds ? edges_lds : 0;
            const int small_lds = (!no_small && !expect_cdf) ? 1 : 0;
            if (small_lds && dyn < small_bytes) dyn = small_bytes;
            hipLaunchKernelGGL((d <= 4 ? k_bucket_redraw<4> : k_bucket_redraw<QSMC_MAX_D>),
                               dim3(expect_cdf ? 1024 : redraw_blocks), dim3(SCAN_THREADS),
                               dyn, s, model->kind, d,
                               model->min_freq, x_in, ldx_in, n_in, w, inv_norm, offsets, chunks64, h->cdf_scratch, lw,
                               k0, k1, ep, maxiter, x_out, pl, redraw_list, redraw_count, nf, h->gbar + 2,
                               expect_cdf ? 1 : 0, edges_in_lds, small_lds);
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
    const int rc = resample_prefix(h, w, n_in, norm, n_out, seed, epoch, s, false);
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

int qsmc_lw_fuse_canonicalize(qsmc_handle_t h, const double *basis, int32_t dim, int32_t basis_kind,
                              int32_t allow_subnormalized) {
    if (!h) return QSMC_ERR_INVALID;
    h->canon_next.kind = 0;
    if (dim != 4) return QSMC_ERR_UNSUPPORTED;                       // (the fused pass exists for 2 qubits, d = 16)
    if (basis_kind != QSMC_BASIS_DENSE && basis_kind != QSMC_BASIS_PAULI) return QSMC_ERR_INVALID;
    if (basis_kind == QSMC_BASIS_DENSE && !basis) return QSMC_ERR_INVALID;
    h->canon_next.kind = basis_kind == QSMC_BASIS_PAULI ? 1 : 2;
    h->canon_next.allow_sub = allow_subnormalized ? 1 : 0;
    h->canon_next.basis = basis_kind == QSMC_BASIS_PAULI ? nullptr : basis;
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
    CanonSpec canon{0, 0, nullptr};
    double expect = 0.0;
    if (h && h->canon_next.kind) {                                  // one-shot (qsmc_lw_fuse_canonicalize)
        canon = CanonSpec{h->canon_next.kind, h->canon_next.allow_sub, h->canon_next.basis};
        h->canon_next.kind = 0;
    }
    if (h) {                                                        // one-shot (qsmc_lw_expect_redraws)
        expect = h->expect_next;
        h->expect_next = 0.0;
    }
    return resample_philox_impl(h, model, postselect, x_in, ldx_in, n_in, d, w, norm, a, mean, S, n_out, seed,
                                epoch, maxiter, x_out, pl, n_failed_host, stream, canon, RS_STAGE_ALL, expect);
}

int qsmc_lw_expect_redraws(qsmc_handle_t h, int64_t n_expected) {
    if (!h || n_expected < 0) return QSMC_ERR_INVALID;
    h->expect_next = (double)n_expected;
    return QSMC_OK;
}

int qsmc_step_stats(qsmc_handle_t h, int64_t *n_queued, int64_t *n_adopted) {
    if (!h || !n_queued || !n_adopted) return QSMC_ERR_INVALID;
    *n_queued = h->rsq.n_queued;
    *n_adopted = h->rsq.n_adopted;
    return QSMC_OK;
}

int qsmc_step_adopted(qsmc_handle_t h) {
    if (!h) return QSMC_ERR_INVALID;
    ++h->rsq.n_adopted;
    return QSMC_OK;
}

int qsmc_step_sqrt_stats(qsmc_handle_t h, int64_t *n_device, int64_t *n_agreed) {
    if (!h || !n_device || !n_agreed) return QSMC_ERR_INVALID;
    *n_device = h->n_sqrt_dev;
    *n_agreed = h->n_sqrt_agreed;
    return QSMC_OK;
}

int qsmc_lw_can_fuse_canonicalize(int32_t d, int64_t n_in, int64_t n_out) {
    if (d != 16 || n_in <= 0 || n_out <= 0) return 0;
    return use_buckets((n_in + BUCKET_CHUNK - 1) / BUCKET_CHUNK, n_out) ? 1 : 0;
}

int qsmc_reserve(qsmc_handle_t h, int64_t n_in, int64_t n_out, int32_t d) {
    if (!h || n_in <= 0 || n_out <= 0 || d < 1 || d > QSMC_MAX_D_WIDE) return QSMC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    // every buffer the update and a resample of this shape grow on first use, grown now
    const int per_block = QSMC_BLOCK * 2 * UPD_UNROLL;
    const int dmom = d <= 4 ? d : 0;
    int rc = ensure_partials(h, (size_t)QSMC_GRID_CAP * (3 + dmom + dmom * (dmom + 1) / 2 + 1));
    if (rc) return rc;
    const int64_t chunks64 = (n_in + BUCKET_CHUNK - 1) / BUCKET_CHUNK;
    rc = ensure_tile_sums(h, (size_t)chunks64 * (BUCKET_CHUNK / (per_block / 2)) * QSMC_WAVES_PER_BLOCK);
    if (rc) return rc;
    if (chunks64 <= TILE_PREFIX_MAX_CHUNKS && h->tile_prefix_cap < (size_t)TILE_PREFIX_MAX_CHUNKS + 1) {
        if (h->tile_prefix) HIP_TRY(h, hipFree(h->tile_prefix));
        h->tile_prefix = nullptr;
        h->tile_prefix_cap = 0;
        HIP_TRY(h, hipMalloc(&h->tile_prefix, (size_t)(TILE_PREFIX_MAX_CHUNKS + 1) * sizeof(double)));
        h->tile_prefix_cap = TILE_PREFIX_MAX_CHUNKS + 1;
    }
    rc = ensure_rs_offsets(h, (size_t)chunks64 + 1);
    if (rc) return rc;
    const bool split16 = qsmc_lw_can_fuse_canonicalize(d, n_in, n_out) != 0;
    BucketPlan bp;
    rc = bucket_plan_layout(h, chunks64, n_out, &bp, split16 || d > QSMC_MAX_D);
    if (rc) return rc;
    if (d > QSMC_MAX_D) {
        const int KW = wide_mom_k((d + 15) / 16);
        rc = ensure_partials(h, (size_t)768 * KW);
        if (rc) return rc;
        rc = ensure_scratch(h, 256 + (size_t)KW);
        if (rc) return rc;
        rc = ensure_anc16(h, ((size_t)(n_out > n_in ? n_out : n_in) + 16) * sizeof(unsigned int));
        if (rc) return rc;
        if (!bp.bucketed) {
            rc = ensure_cdf(h, (size_t)n_in);
            if (rc) return rc;
        }
        rc = ensure_pinned(h, (size_t)KW);
        if (rc) return rc;
    } else if (!split16) {
        rc = ensure_cdf(h, (size_t)n_in);
        if (rc) return rc;
    } else {
        const int gridm = grid_for(n_in, QSMC_BLOCK) < 1024 ? grid_for(n_in, QSMC_BLOCK) : 1024;
        rc = ensure_partials(h, (size_t)gridm * MFMA_MOM_K);
        if (rc) return rc;
        rc = ensure_scratch(h, 256 + MFMA_MOM_K);
        if (rc) return rc;
    }
    return ensure_pinned(h, 64);
}

// mean = S1 / norm, cov = S2 / norm - mean mean^T from the packed sums of the fused update: the operations
// ParticleDistribution._moments / _cov_from_sums perform (one division, one product, one subtraction per entry; this
// file is compiled with -ffp-contract=off), so the caller's own numbers come out bit for bit
static void moments_to_mean_cov(const double *packed, int d, double norm, double *mean, double *cov) {
    for (int m = 0; m < d; ++m) mean[m] = packed[m] / norm;
    int k = d;
    for (int m = 0; m < d; ++m)
        for (int q = m; q < d; ++q) {
            const double e2 = packed[k++] / norm;
            cov[m * d + q] = e2 - mean[m] * mean[q];
            cov[q * d + m] = e2 - mean[q] * mean[m];
        }
}

int qsmc_step(qsmc_handle_t h, qsmc_step_t *st, const qsmc_model_t *model, const qsmc_expparam_t *exp, int64_t outcome,
              qsmc_stream_t stream) {
    if (!h || !st || !model || !exp || !st->x || !st->w_alt || st->n <= 0) return QSMC_ERR_INVALID;
    st->status = 0;
    const int d = model->d;
    const bool small_d = d >= 1 && d <= 4;
    if (st->lw.prefix && st->check_for_resample) {           // (qsmc_lw_arm_prefix, from the struct)
        h->spec.enabled = st->lw.n_out > 0;
        h->spec.thresh = st->ess_below;
        h->spec.n_out = st->lw.n_out;
        h->spec.seed = st->lw.seed;
        h->spec.epoch = st->lw.epoch;
    } else {
        h->spec.enabled = 0;
    }
    int rc = qsmc_update_fused(h, model, st->x, st->ldx, st->n, st->w, st->w_alt, st->norm, exp, outcome, nullptr,
                               &st->stats, small_d ? st->moments : nullptr, stream);
    if (rc) return rc;
    st->update_token = h->ts.gen;
    if (st->ex_segment) {
        // a shard: the one per-datum collective of the sharded updater, here instead of a trip through the caller
        if (st->ex_world < 1 || st->ex_world > QSMC_STEP_MAX_RANKS || !st->ex_k || st->lw.enabled || st->lw.prefix)
            return QSMC_ERR_INVALID;
        const int n_mom = small_d ? d + d * (d + 1) / 2 : 0, nv = 4 + n_mom;
        double vec[4 + 14], tot[4 + 14], rows[QSMC_STEP_MAX_RANKS * (4 + 14)];
        vec[0] = st->stats.sum;
        vec[1] = st->stats.sumsq;
        vec[2] = st->stats.min;
        vec[3] = st->stats.n_bad;
        for (int k = 0; k < n_mom; ++k) vec[4 + k] = st->moments[k];
        rc = qsmc_host_allreduce(st->ex_segment, st->ex_rank, st->ex_world, st->ex_max_len, ++*st->ex_k, vec, nv, 2, rows, tot,
                                 st->ex_timeout_s);
        if (rc) return rc;
        st->stats.sum = tot[0];
        st->stats.sumsq = tot[1];
        st->stats.min = tot[2];
        st->stats.n_bad = tot[3];
        for (int k = 0; k < n_mom; ++k) st->moments[k] = tot[4 + k];
        for (int r = 0; r < st->ex_world; ++r) st->shard_sums[r] = rows[(size_t)r * nv];
    }
    if (st->lw.redraw_pending) {                               // this update's reduction published the last resample's count
        st->lw.redraws_seen = (int64_t)h->mapped[REDUCE_OUT_MAX - 2];
        st->lw.redraw_pending = 0;
    }
    const double norm = st->stats.sum;
    const double fixed = fabs(norm) < PREFIX_NORM_EPS ? 1.0 : norm;                       // smc.py:369-370
    if (st->stats.n_bad > 0.0) { st->status = QSMC_STEP_GUARD; return QSMC_OK; }       // smc.py:416-418
    const double sum_w = norm / fixed;
    if (sum_w <= st->zero_weight_thresh || !(sum_w == sum_w)) { st->status = QSMC_STEP_GUARD; return QSMC_OK; }   // :423-436
    // commit (smc.py:441): the new weights become the cloud's, the old buffer the next scratch
    const double *old_w = st->w;
    st->w = st->w_alt;
    st->w_alt = const_cast<double *>(old_w);
    st->norm = fixed;
    st->sumsq = st->stats.sumsq;
    const double n2 = fixed * fixed;
    const double ess = st->stats.sumsq == 0.0 ? (n2 > 0.0 ? INFINITY : NAN) : n2 / st->stats.sumsq;
    st->n_ess = ess;
    if (ess <= st->min_n_ess) st->min_n_ess = ess;
    if (!st->check_for_resample) return QSMC_OK;
    if (ess <= 10.0) st->status |= QSMC_STEP_SMALL_ESS;
    if (!(ess < st->ess_below)) return QSMC_OK;
    st->status |= QSMC_STEP_RESAMPLE_DUE;
    constexpr bool no_queue = false;
    if (st->ex_segment && st->plan_enabled && !no_queue) {
        // a shard: plan the resample (every rank draws the same plan from the shard sums it has just received) and start
        // this shard's weight-only prefix; mean / covariance / square root and the sampler are the caller's, behind it
        const int G = st->ex_world;
        rc = qsmc_shard_plan_totals(st->plan_seed, st->plan_epoch, st->shard_sums, G, st->plan_n_total, st->plan_totals);
        if (rc) return QSMC_OK;                                // (odd shard sums: the caller's own plan raises)
        st->status |= QSMC_STEP_PLAN_READY;
        const int64_t target = st->plan_n_total / G;
        int64_t dev = 0, tmin = st->plan_totals[0];
        for (int r = 0; r < G; ++r) {
            const int64_t e = st->plan_totals[r] > target ? st->plan_totals[r] - target : target - st->plan_totals[r];
            dev = e > dev ? e : dev;
            tmin = st->plan_totals[r] < tmin ? st->plan_totals[r] : tmin;
        }
        const double drift = (double)dev / (double)(target > 1 ? target : 1);
        st->plan_stay = (drift <= st->plan_tol && tmin > 0) ? 1 : 0;
        if (st->plan_stay && st->shard_sums[st->ex_rank] > 0.0) {
            h->ts.armed = h->ts.gen;                           // these weights ARE update number ts.gen's output
            rc = qsmc_lw_resample_prepare(h, st->w, st->n, st->shard_sums[st->ex_rank], st->plan_totals[st->ex_rank],
                                          st->plan_prefix_seed, st->plan_epoch, stream);
            if (rc) return rc;
            st->status |= QSMC_STEP_PREFIX_QUEUED;
        }
        return QSMC_OK;
    }
    if (!st->lw.enabled || !st->lw.x_out || st->lw.n_out <= 0 || no_queue) return QSMC_OK;
    const CanonSpec canon{st->lw.canon_kind, st->lw.canon_allow_sub, st->lw.canon_basis};
    hipStream_t s = (hipStream_t)stream;
    OutPlace pl;
    memset(&pl, 0, sizeof(pl));
    pl.ld_m = st->lw.ldx_out;
    pl.ld_s = 1;
    bool device_sqrt = false;
    if (small_d) {
        if (canon.kind) return QSMC_OK;
        // the caller's resample (resamplers.py:266-300), started from here
        moments_to_mean_cov(st->moments, d, fixed, st->mean, st->cov);
    } else {
        // d = 16 tomography on the split sampler: moments (their own pass) and the ancestors are queued now; the host
        // forms S while the ancestor kernel runs and queues the kicks behind it
        const int64_t chunks64 = (st->n + BUCKET_CHUNK - 1) / BUCKET_CHUNK;
        if (d != 16 || model->kind != QSMC_MODEL_TOMOGRAPHY || !use_buckets(chunks64, st->lw.n_out))
            return QSMC_OK;
        const int gridm = grid_for(st->n, QSMC_BLOCK) < 1024 ? grid_for(st->n, QSMC_BLOCK) : 1024;
        rc = ensure_partials(h, (size_t)gridm * MFMA_MOM_K);
        if (rc) return rc;
        rc = ensure_scratch(h, 256 + MFMA_MOM_K);
        if (rc) return rc;
        hipEvent_t m0 = nullptr, m1 = nullptr;
        prof_events(h, QSMC_PROF_MOMENTS, &m0, &m1);
        hipExtLaunchKernelGGL(k_moments_mfma, dim3(gridm), dim3(QSMC_BLOCK), 0, s, m0, m1, 0, st->x, st->ldx, st->n, d,
                              st->w, fixed, h->partials);
        double *full = h->scratch + 256;
        const unsigned long long seq = ++h->seq;
        static const bool dev_sqrt_env = getenv("QSMC_DEVICE_SQRT") != nullptr;
        constexpr int SUM_GRID = (MFMA_MOM_K + QSMC_WAVES_PER_BLOCK - 1) / QSMC_WAVES_PER_BLOCK;
        // (Round 5 merged these sums with their publish -- every workgroup storing to pinned memory, a ticket, the last one
        //  setting the completion word -- and measured it: 13.8-14.5 us against 7.0 + 4.7 for the pair, whether the ticket is a
        //  release or relaxed and whether every thread or only the writing lanes fence at system scope: 273 pinned stores from
        //  69 workgroups each wait for their own acknowledgement, where k_publish_big's one workgroup waits once.  Like the
        //  one-launch datum: on this part a dependent launch whose packet is already queued costs about what
        //  any in-kernel hand-over does.  The merged kernel was removed in round 6.)
        hipLaunchKernelGGL(k_sum_partials, dim3(SUM_GRID), dim3(QSMC_BLOCK), 0, s, h->partials, gridm, MFMA_MOM_K, full);
        // (round 4 built the device form -- kernels/sqrtm.hpp -- and measured it: the gap between the two sampler kernels
        //  closes, but the one wavefront that forms S is latency-bound, ~90 rounds of three dependent LDS / fp64-division
        //  steps sharing a SIMD with the ancestor kernel's own waves: k_bucket_anc16 36 -> 96 us, a d = 16 resample
        //  +40 us, config-5 share 0.0727 -> 0.0742 ms/step.  The host's Jacobi (~25 us) runs WHILE the ancestor kernel
        //  does and leaves a 10-25 us gap: it stays the default; QSMC_DEVICE_SQRT=1 selects the device form.)
        device_sqrt = dev_sqrt_env;
        h->ts.armed = h->ts.gen;                               // these weights ARE update number ts.gen's output
        if (device_sqrt) {
            // round 4: nothing of the resample waits for the host any more.  One wavefront riding in the ancestor kernel
            // (kernels/sqrtm.hpp) turns the summed moments into mean / covariance / S = h sqrtm_psd(cov) in device memory,
            // the kick kernel reads them from there, canonicalize's list pass follows: all queued here, now.  The host then
            // receives moments, S and its error, repeats the square root with the same routine and adopts the queued
            // resample only if every bit agrees (else the caller's own call runs it again with the host's numbers).
            SqrtJob sj;
            sj.full = full;
            sj.out = h->lw_dev;
            sj.mapped = h->mapped_big_dev;
            sj.flag = h->flag_dev;
            sj.seq = seq;
            sj.a = st->lw.a;
            sj.h = st->lw.h;
            sj.zero_cov_comp = st->lw.zero_cov_comp;
            HIP_TRY(h, hipGetLastError());
            rc = resample_philox_impl(h, model, st->lw.postselect, st->x, st->ldx, st->n, d, st->w, fixed, st->lw.a, st->mean,
                                      st->S, st->lw.n_out, st->lw.seed, st->lw.epoch, st->lw.maxiter, st->lw.x_out, pl,
                                      nullptr, stream, canon, RS_STAGE_ALL, 0.0, &sj, h->lw_dev);
            if (rc) return rc;
            ++h->n_sqrt_dev;
        } else {
            hipLaunchKernelGGL(k_publish_big, dim3(1), dim3(QSMC_BLOCK), 0, s, full, MFMA_MOM_K, h->mapped_big_dev, h->flag_dev, seq);
            HIP_TRY(h, hipGetLastError());
            rc = resample_philox_impl(h, model, st->lw.postselect, st->x, st->ldx, st->n, d, st->w, fixed, st->lw.a, st->mean,
                                      st->S, st->lw.n_out, st->lw.seed, st->lw.epoch, st->lw.maxiter, st->lw.x_out, pl,
                                      nullptr, stream, canon, RS_STAGE_ANCESTORS);
            if (rc) return rc;
        }
        rc = wait_reduction(h, s);
        if (rc) return rc;
        // qsmc_moments' packing of the same block: [sum w, sum w x (d), upper(sum w x x^T)], weights already / norm
        const double *hf = h->mapped_big;
        st->moments_big[0] = hf[272];
        int k = 1 + d;
        for (int m = 0; m < d; ++m) {
            st->moments_big[1 + m] = hf[256 + m];
            for (int q = m; q < d; ++q) st->moments_big[k++] = hf[m * 16 + q];
        }
        for (int m = 0; m < d; ++m) st->mean[m] = st->moments_big[1 + m];
        k = 1 + d;
        for (int m = 0; m < d; ++m)
            for (int q = m; q < d; ++q) {
                const double e2 = st->moments_big[k++];
                st->cov[m * d + q] = e2 - st->mean[m] * st->mean[q];
                st->cov[q * d + m] = e2 - st->mean[q] * st->mean[m];
            }
    }
    bool finite = true, any = false;
    for (int k = 0; k < d * d; ++k) {
        finite = finite && std::isfinite(st->cov[k]);
        any = any || st->cov[k] != 0.0;
    }
    if (!finite) return QSMC_OK;                               // (the caller's own assertion fires)
    double cov_used[QSMC_MAX_D * QSMC_MAX_D];
    for (int k = 0; k < d * d; ++k) cov_used[k] = any ? st->cov[k] : ((k / d == k % d) ? st->lw.zero_cov_comp : 0.0);
    sqrtm_psd_host(cov_used, d, st->lw.h, st->S, &st->S_err, &st->cov_lambda_min);
    if (!std::isfinite(st->S_err)) return QSMC_OK;             // (ResamplerError is the caller's to raise)
    if (small_d) h->ts.armed = h->ts.gen;                      // these weights ARE update number ts.gen's output
    const double expect = small_d ? (double)st->lw.redraws_seen : 0.0;
    if (device_sqrt) {
        // the device's square root against the host's: same routine, same input -- the same bits, or no adoption
        const double *dv = h->mapped_big;
        if (dv[SQRT_MAPPED_VALID] != 1.0 || memcmp(&dv[SQRT_MAPPED_ERR], &st->S_err, sizeof(double)) != 0 ||
            memcmp(dv + SQRT_MAPPED_S, st->S, sizeof(double) * d * d) != 0)
            return QSMC_OK;                                    // (the caller's own call repeats the resample with its S)
        ++h->n_sqrt_agreed;
    } else {
        rc = resample_philox_impl(h, model, st->lw.postselect, st->x, st->ldx, st->n, d, st->w, fixed, st->lw.a, st->mean,
                                  st->S, st->lw.n_out, st->lw.seed, st->lw.epoch, st->lw.maxiter, st->lw.x_out, pl, nullptr,
                                  stream, canon, small_d ? RS_STAGE_ALL : RS_STAGE_KICK, expect);
        if (rc) return rc;
    }
    st->lw.redraw_pending = 1;
    auto &q = h->rsq;
    q.valid = 1;
    q.expect = expect;
    q.canon_kind = canon.kind;
    q.canon_allow_sub = canon.allow_sub;
    q.canon_basis = canon.basis;
    q.model = *model;
    q.postselect = st->lw.postselect;
    q.d = d;
    q.maxiter = st->lw.maxiter;
    q.x_in = st->x;
    q.w = st->w;
    q.ldx_in = st->ldx;
    q.n_in = st->n;
    q.n_out = st->lw.n_out;
    q.ldx_out = st->lw.ldx_out;
    q.norm = fixed;
    q.a = st->lw.a;
    memcpy(q.mean, st->mean, sizeof(double) * d);
    memcpy(q.S, st->S, sizeof(double) * d * d);
    q.seed = st->lw.seed;
    q.epoch = st->lw.epoch;
    q.x_out = st->lw.x_out;
    q.stream = s;
    ++q.n_queued;
    if (st->lw.adopt) {
        // the caller MAY take the queued resample as its own, without a second call to qsmc_lw_resample_philox: it says so
        // with qsmc_step_adopted (counted there, not here -- a caller whose resampler was edited in place since the struct
        // was filled runs its own resample and must not show up as an adoption); the record is closed either way
        q.valid = 0;
    }
    st->status |= QSMC_STEP_RESAMPLE_QUEUED;
    return QSMC_OK;
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
    if (h) h->expect_next = 0.0;                       // (the proposal bank is not used by the sharded form)
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

int qsmc_last_resample_redraws(qsmc_handle_t h, int64_t *n_redraws_out) {
    if (!h || !n_redraws_out) return QSMC_ERR_INVALID;
    *n_redraws_out = (int64_t)h->mapped[REDUCE_OUT_MAX - 2];
    return QSMC_OK;
}

int qsmc_prior_uniform_philox(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect,
                              const double *lo, const double *hi, int32_t d, int64_t n, uint64_t seed,
                              uint64_t epoch, int32_t maxiter, double *x_out, int64_t ldx_out,
                              int64_t *n_failed_host, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->rsq.valid = 0; h->canon_next.kind = 0; h->expect_next = 0.0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
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
    if (!h || !x || !scale || n < 0 || d < 1 || d > QSMC_MAX_D_WIDE) return QSMC_ERR_INVALID;
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

int qsmc_tomo_canonicalize2(qsmc_handle_t h, const double *basis, int32_t dim, int32_t basis_kind, double *x, int64_t ldx,
                            int64_t n, int32_t allow_subnormalized, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->rsq.valid = 0; h->canon_next.kind = 0; h->expect_next = 0.0; h->ts.w = nullptr; }   // weights / counters are about to change: drop queued resample state
    if (!h || !x || n < 0) return QSMC_ERR_INVALID;
    if (basis_kind != QSMC_BASIS_DENSE && basis_kind != QSMC_BASIS_PAULI) return QSMC_ERR_INVALID;
    if (basis_kind == QSMC_BASIS_DENSE && !basis) return QSMC_ERR_INVALID;
    if (n == 0) return QSMC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n, QSMC_BLOCK);
    switch (dim) {
        case 2:
            if (!basis) return QSMC_ERR_INVALID;        // (one qubit: the dense contraction is 4 x 4 already)
            hipLaunchKernelGGL((k_tomo_canon<2, TomoDense<2>>), dim3(grid), dim3(QSMC_BLOCK), 0, s, TomoDense<2>{basis}, x, ldx,
                               n, allow_subnormalized);
            break;
        case 3:                                         // a qutrit (d = 9): one pass, dense contraction, 3 x 3 Jacobi per lane
            if (!basis) return QSMC_ERR_INVALID;
            hipLaunchKernelGGL((k_tomo_canon<3, TomoDense<3>>), dim3(grid), dim3(QSMC_BLOCK), 0, s, TomoDense<3>{basis}, x, ldx,
                               n, allow_subnormalized);
            break;
        case 4:
            if (basis_kind == QSMC_BASIS_PAULI) return canon_dim4(h, TomoPauli2{}, x, ldx, n, allow_subnormalized, s);
            return canon_dim4(h, TomoDense<4>{basis}, x, ldx, n, allow_subnormalized, s);
        case 5: return canon_wide<5>(h, basis, x, ldx, n, allow_subnormalized, s);
        case 6: return canon_wide<6>(h, basis, x, ldx, n, allow_subnormalized, s);
        case 7: return canon_wide<7>(h, basis, x, ldx, n, allow_subnormalized, s);
        case 8: return canon_wide<8>(h, basis, x, ldx, n, allow_subnormalized, s);   // three qubits
        default: return QSMC_ERR_UNSUPPORTED;
    }
    HIP_TRY(h, hipGetLastError());
    return QSMC_OK;
}

int qsmc_tomo_canonicalize(qsmc_handle_t h, const double *basis, int32_t dim, double *x, int64_t ldx,
                           int64_t n, int32_t allow_subnormalized, qsmc_stream_t stream) {
    return qsmc_tomo_canonicalize2(h, basis, dim, QSMC_BASIS_DENSE, x, ldx, n, allow_subnormalized, stream);
}

// ---- posterior read-outs: sort by weight / by location, search a sorted table -------
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
    // kernels/sort.hpp: eight 8-bit passes ping-ponging between a temporary pair and the output pair (which holds key
    // images until the last pass writes doubles); pass 0 reads the caller's keys, index = position
    const int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    const long long m = 256ll * ntiles;
    const int nchunks = (int)((m + SORT_SCAN_CHUNK - 1) / SORT_SCAN_CHUNK);
    const size_t pair_bytes = (((size_t)n * 8) + 255) & ~(size_t)255;
    const size_t hist_bytes = (((size_t)m * sizeof(unsigned int)) + 255) & ~(size_t)255;
    const size_t need = 2 * pair_bytes + hist_bytes + (size_t)nchunks * sizeof(unsigned int) + 256;
    if (h->sort_tmp_cap < need) {
        if (h->sort_tmp) HIP_TRY(h, hipFree(h->sort_tmp));
        h->sort_tmp = nullptr;
        h->sort_tmp_cap = 0;
        HIP_TRY(h, hipMalloc(&h->sort_tmp, need));
        h->sort_tmp_cap = need;
    }
    char *base = static_cast<char *>(h->sort_tmp);
    void *tk = base;
    long long *ti = reinterpret_cast<long long *>(base + pair_bytes);
    unsigned int *hist = reinterpret_cast<unsigned int *>(base + 2 * pair_bytes);
    unsigned int *totals = reinterpret_cast<unsigned int *>(base + 2 * pair_bytes + hist_bytes);
    const int desc = descending ? 1 : 0;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 8 * pass;
        const void *kin = pass == 0 ? (const void *)keys : ((pass & 1) ? (const void *)tk : (const void *)keys_out);
        const long long *iin = pass == 0 ? nullptr : ((pass & 1) ? ti : (const long long *)idx_out);
        void *kout = (pass & 1) ? (void *)keys_out : tk;
        long long *iout = (pass & 1) ? (long long *)idx_out : ti;
        if (pass == 0)
            hipLaunchKernelGGL((k_sort_hist<true>), dim3(ntiles), dim3(QSMC_BLOCK), 0, s, kin, (long long)n, shift, desc, hist, ntiles);
        else
            hipLaunchKernelGGL((k_sort_hist<false>), dim3(ntiles), dim3(QSMC_BLOCK), 0, s, kin, (long long)n, shift, desc, hist, ntiles);
        hipLaunchKernelGGL(k_sort_scan, dim3(nchunks), dim3(1024), 0, s, hist, m, totals);
        hipLaunchKernelGGL(k_sort_scan_top, dim3(1), dim3(1024), 0, s, totals, nchunks);
        if (pass == 0)
            hipLaunchKernelGGL((k_sort_scatter<true, false>), dim3(ntiles), dim3(QSMC_BLOCK), 0, s, kin, iin, kout, iout,
                               (long long)n, shift, desc, hist, totals, ntiles);
        else if (pass == 7)
            hipLaunchKernelGGL((k_sort_scatter<false, true>), dim3(ntiles), dim3(QSMC_BLOCK), 0, s, kin, iin, kout, iout,
                               (long long)n, shift, desc, hist, totals, ntiles);
        else
            hipLaunchKernelGGL((k_sort_scatter<false, false>), dim3(ntiles), dim3(QSMC_BLOCK), 0, s, kin, iin, kout, iout,
                               (long long)n, shift, desc, hist, totals, ntiles);
    }
    HIP_TRY(h, hipGetLastError());
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

// ---- RCCL: the per-datum reduction of the sharded updater on the launch stream ----------
// Replaces what the reference does with an ipyparallel gather of the likelihood array (parallel.py:216-224): here
// only the update kernel's sums cross the links.  ONE collective on `stream`, right behind the kernel that produced
// the vector: an all-gather of every rank's n doubles (18 at d = 3: bytes, not bandwidth); a one-workgroup kernel then
// sums the rows in rank order (the weight minimum: min; entry 0 of every row doubles as that shard's weight total, the
// next resample's shard plan) and publishes the result in the pinned slot the completion word already guards, so the
// host's only wait per datum is the same spin it does on one GPU.  Round 2 used all-reduce(sum) + all-reduce(min) +
// all-gather in a group: three collectives, and the sums' association order was RCCL's; a gather moves bits only, so
// every rank -- and the shared-memory transport -- forms the same totals and takes the same resample decision.  librccl is bound with dlsym, preferring the copy already in the process
// (torch's): two RCCLs would mean two HIP runtimes.
constexpr int QSMC_MAX_RANKS = 64;

static void *rccl_open() {
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names)
        if (void *l = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) return l;
    for (const char *n : names)
        if (void *l = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) return l;
    return nullptr;
}

int qsmc_comm_unique_id(void *id_out) {
    if (!id_out) return QSMC_ERR_INVALID;
    void *lib = rccl_open();
    if (!lib) return QSMC_ERR_UNSUPPORTED;
    auto get = reinterpret_cast<decltype(&ncclGetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    if (!get) return QSMC_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (get(&id) != ncclSuccess) return QSMC_ERR_HIP;
    memcpy(id_out, &id, sizeof(id));
    return QSMC_OK;
}

int qsmc_comm_init(qsmc_handle_t h, int32_t rank, int32_t nranks, const void *unique_id) {
    if (!h || !unique_id || nranks < 1 || nranks > QSMC_MAX_RANKS || rank < 0 || rank >= nranks) return QSMC_ERR_INVALID;
    if (h->cc.comm) return QSMC_ERR_INVALID;
    void *lib = rccl_open();
    if (!lib) {
        snprintf(h->hip_err, sizeof(h->hip_err), "librccl not found: %s", dlerror());
        return QSMC_ERR_UNSUPPORTED;
    }
    h->cc.lib = lib;
#define BIND(NAME)                                                                         \
    h->cc.NAME = reinterpret_cast<decltype(&nccl##NAME)>(dlsym(lib, "nccl" #NAME));         \
    if (!h->cc.NAME) { snprintf(h->hip_err, sizeof(h->hip_err), "nccl" #NAME " missing"); return QSMC_ERR_UNSUPPORTED; }
    BIND(CommInitRank) BIND(CommDestroy) BIND(CommCount) BIND(CommUserRank) BIND(AllGather) BIND(GetErrorString)
#undef BIND
    HIP_TRY(h, hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = h->cc.CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) {
        snprintf(h->hip_err, sizeof(h->hip_err), "ncclCommInitRank: %s", h->cc.GetErrorString(r));
        return QSMC_ERR_HIP;
    }
    HIP_TRY(h, hipMalloc(&h->cc.buf, (size_t)QSMC_MAX_RANKS * REDUCE_OUT_MAX * sizeof(double)));
    h->cc.comm = comm;
    h->cc.rank = rank;
    h->cc.nranks = nranks;
    return QSMC_OK;
}

int qsmc_comm_destroy(qsmc_handle_t h) {
    if (!h) return QSMC_ERR_INVALID;
    if (h->cc.comm) {
        (void)h->cc.CommDestroy(h->cc.comm);
        h->cc.comm = nullptr;
    }
    if (h->cc.buf) {
        (void)hipFree(h->cc.buf);
        h->cc.buf = nullptr;
    }
    return QSMC_OK;
}

// rows[r][0..n): rank r's vector as the all-gather delivered it.  Thread t sums entry t over the ranks IN RANK ORDER
// (entry min_index: the minimum, NaN propagating) -- the very loop qsmc_host_allreduce runs on the host, so the two
// transports agree bit for bit and, an all-gather moving bits without arithmetic, so do all ranks.
__global__ void k_publish_allgather(const double *__restrict__ rows, int n, int min_index, int nranks,
                                    double *__restrict__ mapped, unsigned long long *flag, unsigned long long seq,
                                    const unsigned long long *__restrict__ counters, double *__restrict__ failed_dst) {
    const int t = threadIdx.x;
    if (t < n) {
        double acc = rows[t];
        if (t == min_index) {
            for (int r = 1; r < nranks; ++r) {
                const double v = rows[(size_t)r * n + t];
                acc = (v < acc || v != v) ? v : acc;
            }
        } else {
            for (int r = 1; r < nranks; ++r) acc += rows[(size_t)r * n + t];
        }
        mapped[t] = acc;
    }
    if (t < nranks) mapped[n + t] = rows[(size_t)t * n];
    if (t == 0) {                    // like every host-visible reduction: the last resample's failed / redraw counts ride along
        failed_dst[0] = (double)counters[0];
        failed_dst[-1] = (double)counters[1];
    }
    __syncthreads();
    if (t == 0) {
        __threadfence_system();
        *reinterpret_cast<volatile unsigned long long *>(flag) = seq;
    }
}

int qsmc_comm_count(qsmc_handle_t h, int32_t *nranks_out, int32_t *rank_out) {
    if (!h || !nranks_out) return QSMC_ERR_INVALID;
    if (!h->cc.comm) return QSMC_ERR_INVALID;
    int count = -1, rank = -1;
    ncclResult_t r = h->cc.CommCount(h->cc.comm, &count);
    if (r == ncclSuccess) r = h->cc.CommUserRank(h->cc.comm, &rank);
    if (r != ncclSuccess) {
        snprintf(h->hip_err, sizeof(h->hip_err), "rccl: %s", h->cc.GetErrorString(r));
        return QSMC_ERR_HIP;
    }
    *nranks_out = count;
    if (rank_out) *rank_out = rank;
    return QSMC_OK;
}

int qsmc_allreduce_sums(qsmc_handle_t h, const double *vec_dev, int32_t n, int32_t min_index, double *tot_host,
                        double *firsts_host, qsmc_stream_t stream) {
    if (!h || !vec_dev || !tot_host || n < 1 || min_index >= n) return QSMC_ERR_INVALID;
    if (!h->cc.comm) return QSMC_ERR_INVALID;
    // the result and every rank's entry 0 share the pinned block with the four reserved tail slots (failed count,
    // redraw count, prefix gate and its normaliser)
    if (n + h->cc.nranks > REDUCE_OUT_MAX - 4) return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const ncclResult_t r = h->cc.AllGather(vec_dev, h->cc.buf, (size_t)n, ncclDouble, h->cc.comm, s);
    if (r != ncclSuccess) {
        snprintf(h->hip_err, sizeof(h->hip_err), "rccl: %s", h->cc.GetErrorString(r));
        return QSMC_ERR_HIP;
    }
    const unsigned long long seq = ++h->seq;
    hipLaunchKernelGGL(k_publish_allgather, dim3(1), dim3(256), 0, s, h->cc.buf, (int)n, (int)min_index, h->cc.nranks,
                       h->mapped_dev, h->flag_dev, seq, reinterpret_cast<const unsigned long long *>(h->counter),
                       h->mapped_dev + (REDUCE_OUT_MAX - 1));
    HIP_TRY(h, hipGetLastError());
    // a collective can take arbitrarily long when a peer is late: no 20 ms spin window here, wait for the stream
    volatile unsigned long long *f = h->flag;
    for (unsigned spins = 0; *f != seq; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xfffff) == 0xfffff && hipStreamQuery(s) != hipErrorNotReady) {
            HIP_TRY(h, hipStreamSynchronize(s));
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    memcpy(tot_host, h->mapped, (size_t)n * sizeof(double));
    if (firsts_host) memcpy(firsts_host, h->mapped + n, (size_t)h->cc.nranks * sizeof(double));
    return QSMC_OK;
}

// The second half of qsmc_allreduce_sums on rows the caller gathered by other means: rows_dev holds nranks vectors of n
// doubles, rank after rank; the same one-wave kernel sums them IN RANK ORDER (entry min_index: the minimum), keeps every
// rank's entry 0 and publishes through the pinned block.  No communicator needed -- which also makes the device half of
// the RCCL transport testable with any number of "ranks" on one GPU (tests/test_gpu_parity.py).
int qsmc_publish_rows(qsmc_handle_t h, const double *rows_dev, int32_t n, int32_t min_index, int32_t nranks, double *tot_host,
                      double *firsts_host, qsmc_stream_t stream) {
    if (!h || !rows_dev || !tot_host || n < 1 || min_index >= n || nranks < 1 || nranks > QSMC_MAX_RANKS) return QSMC_ERR_INVALID;
    if (n + nranks > REDUCE_OUT_MAX - 4) return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const unsigned long long seq = ++h->seq;
    hipLaunchKernelGGL(k_publish_allgather, dim3(1), dim3(256), 0, s, rows_dev, (int)n, (int)min_index, (int)nranks,
                       h->mapped_dev, h->flag_dev, seq, reinterpret_cast<const unsigned long long *>(h->counter),
                       h->mapped_dev + (REDUCE_OUT_MAX - 1));
    HIP_TRY(h, hipGetLastError());
    const int rc = wait_reduction(h, s);
    if (rc) return rc;
    memcpy(tot_host, h->mapped, (size_t)n * sizeof(double));
    if (firsts_host) memcpy(firsts_host, h->mapped + n, (size_t)nranks * sizeof(double));
    return QSMC_OK;
}

// ---- user models compiled at run time (kernels/user_jit.hpp) ---------------------------------------------------------------
// hiprtc is bound with dlopen: the copy already in the process if there is one, else the path the caller names (the Python
// side passes the one that ships with its torch, so that code object and runtime come from one ROCm), else by soname.
struct qsmc_user_kernel {
    hipModule_t mod;
    hipFunction_t upd, upd_multi, lik, valid;
    int d, n_ep, has_valid;
};

constexpr int USER_MAX_EP = 32;

namespace {
struct Hiprtc {
    void *lib = nullptr;
    int (*CreateProgram)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*CompileProgram)(void *, int, const char **) = nullptr;
    int (*GetProgramLogSize)(void *, size_t *) = nullptr;
    int (*GetProgramLog)(void *, char *) = nullptr;
    int (*GetCodeSize)(void *, size_t *) = nullptr;
    int (*GetCode)(void *, char *) = nullptr;
    int (*DestroyProgram)(void **) = nullptr;
};
Hiprtc g_rtc;

bool hiprtc_open(const char *path_hint) {
    if (g_rtc.lib) return true;
    const char *names[] = {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6"};
    void *l = nullptr;
    for (const char *n : names)
        if (!l) l = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (!l && path_hint && *path_hint) l = dlopen(path_hint, RTLD_NOW | RTLD_LOCAL);
    for (const char *n : names)
        if (!l) l = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!l) return false;
#define RTC_BIND(NAME)                                                                       \
    g_rtc.NAME = reinterpret_cast<decltype(g_rtc.NAME)>(dlsym(l, "hiprtc" #NAME));           \
    if (!g_rtc.NAME) return false;
    RTC_BIND(CreateProgram) RTC_BIND(CompileProgram) RTC_BIND(GetProgramLogSize) RTC_BIND(GetProgramLog)
    RTC_BIND(GetCodeSize) RTC_BIND(GetCode) RTC_BIND(DestroyProgram)
#undef RTC_BIND
    g_rtc.lib = l;
    return true;
}
}  // namespace

int qsmc_user_kernel_build(qsmc_handle_t h, const char *user_source, int32_t d, int32_t n_ep, const char *hiprtc_path,
                           qsmc_user_kernel_t *out, char *log_out, int32_t log_cap) {
    if (log_out && log_cap > 0) log_out[0] = 0;
    if (!h || !user_source || !out || d < 1 || d > QSMC_MAX_D || n_ep < 0 || n_ep > USER_MAX_EP) return QSMC_ERR_INVALID;
    *out = nullptr;
    if (!hiprtc_open(hiprtc_path)) {
        snprintf(h->hip_err, sizeof(h->hip_err), "hiprtc not found (libhiprtc.so)");
        return QSMC_ERR_UNSUPPORTED;
    }
    char defs[128];
    snprintf(defs, sizeof(defs), "#define QSMC_D %d\n#define QSMC_NEP %d\n", (int)d, (int)n_ep);
    const size_t len = strlen(defs) + strlen(USER_JIT_PRELUDE) + strlen(user_source) + strlen(USER_JIT_KERNELS) + 8;
    char *src = static_cast<char *>(malloc(len));
    if (!src) return QSMC_ERR_ALLOC;
    snprintf(src, len, "%s%s\n%s\n%s", defs, USER_JIT_PRELUDE, user_source, USER_JIT_KERNELS);
    void *prog = nullptr;
    int rc = g_rtc.CreateProgram(&prog, src, "qsmc_user_model.hip", 0, nullptr, nullptr);
    free(src);
    if (rc != 0 || !prog) return QSMC_ERR_HIP;
    // the device this handle lives on decides the target; -ffp-contract=off as for the library's own kernels (a user model
    // checked against its NumPy twin must not differ by fused multiply-adds the host never made)
    hipDeviceProp_t prop;
    char arch[96] = "--offload-arch=gfx950";
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.gcnArchName[0]) {
        char name[64];
        snprintf(name, sizeof(name), "%s", prop.gcnArchName);
        if (char *colon = strchr(name, ':')) *colon = 0;                      // ("gfx950:sramecc+:xnack-" -> "gfx950")
        snprintf(arch, sizeof(arch), "--offload-arch=%s", name);
    }
    const char *opts[] = {arch, "-O3", "-ffp-contract=off", "-std=c++17"};
    rc = g_rtc.CompileProgram(prog, 4, opts);
    if (log_out && log_cap > 1) {
        size_t ls = 0;
        if (g_rtc.GetProgramLogSize(prog, &ls) == 0 && ls > 1) {
            char *tmp = static_cast<char *>(malloc(ls + 1));
            if (tmp && g_rtc.GetProgramLog(prog, tmp) == 0) {
                tmp[ls] = 0;
                snprintf(log_out, (size_t)log_cap, "%s", tmp);
            }
            free(tmp);
        }
    }
    if (rc != 0) {
        (void)g_rtc.DestroyProgram(&prog);
        snprintf(h->hip_err, sizeof(h->hip_err), "hiprtc: the user model does not compile (see the log)");
        return QSMC_ERR_INVALID;
    }
    size_t cs = 0;
    if (g_rtc.GetCodeSize(prog, &cs) != 0 || cs == 0) { (void)g_rtc.DestroyProgram(&prog); return QSMC_ERR_HIP; }
    char *code = static_cast<char *>(malloc(cs));
    if (!code) { (void)g_rtc.DestroyProgram(&prog); return QSMC_ERR_ALLOC; }
    rc = g_rtc.GetCode(prog, code);
    (void)g_rtc.DestroyProgram(&prog);
    if (rc != 0) { free(code); return QSMC_ERR_HIP; }
    qsmc_user_kernel *uk = new (std::nothrow) qsmc_user_kernel();
    if (!uk) { free(code); return QSMC_ERR_ALLOC; }
    uk->d = d;
    uk->n_ep = n_ep;
    uk->has_valid = strstr(user_source, "QSMC_USER_HAS_VALID") != nullptr;
    hipError_t e = hipModuleLoadData(&uk->mod, code);
    free(code);
    if (e == hipSuccess) e = hipModuleGetFunction(&uk->upd, uk->mod, "qsmc_user_update");
    if (e == hipSuccess) e = hipModuleGetFunction(&uk->upd_multi, uk->mod, "qsmc_user_update_multi");
    if (e == hipSuccess) e = hipModuleGetFunction(&uk->lik, uk->mod, "qsmc_user_likelihood");
    if (e == hipSuccess) e = hipModuleGetFunction(&uk->valid, uk->mod, "qsmc_user_valid");
    if (e != hipSuccess) {
        snprintf(h->hip_err, sizeof(h->hip_err), "loading the compiled user model: %s", hipGetErrorString(e));
        delete uk;
        return QSMC_ERR_HIP;
    }
    *out = uk;
    return QSMC_OK;
}

int qsmc_user_kernel_destroy(qsmc_user_kernel_t uk) {
    if (!uk) return QSMC_OK;
    (void)hipModuleUnload(uk->mod);
    delete uk;
    return QSMC_OK;
}

struct UserEpArg { double v[USER_MAX_EP]; };

// qsmc_update_fused's contract for a compiled user model: same sums, same host-visible results, same completion-word wait
int qsmc_update_user(qsmc_handle_t h, qsmc_user_kernel_t uk, const double *x, int64_t ldx, int64_t n, const double *w_in,
                     double *w_out, double prev_norm, const double *ep, int64_t outcome, double *stats_dev,
                     qsmc_update_stats_t *stats_host, double *moments_host, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->rsq.valid = 0; h->canon_next.kind = 0; h->expect_next = 0.0; h->ts.w = nullptr; }
    if (!h || !uk || !x || !w_out || n <= 0 || (uk->n_ep > 0 && !ep)) return QSMC_ERR_INVALID;
    const int d = uk->d, dmom = d <= 4 ? d : 0;
    if (moments_host && !dmom) return QSMC_ERR_UNSUPPORTED;
    const int n_mom = dmom + dmom * (dmom + 1) / 2, ns = 3 + n_mom;
    hipStream_t s = (hipStream_t)stream;
    const int unroll = d <= 2 ? 4 : (d <= 4 ? 2 : 1);                // (QSMC_JIT_UNROLL of kernels/user_jit.hpp)
    const int grid = grid_for(n, 256 * 2 * unroll);
    int vec = (aligned16(x) && (!w_in || aligned16(w_in)) && aligned16(w_out) && (ldx % 2 == 0)) ? 1 : 0;
    int rc = ensure_partials(h, (size_t)grid * (ns + 1));
    if (rc) return rc;
    ReduceOut ro = make_reduce(h, stats_host || moments_host, stats_dev);
    ++h->ts.gen;                                   // (no tile sums: a resample that follows reads the weights itself)
    h->ts.armed = 0;
    h->spec.launched = 0;
    // the kernel's struct argument is sized by the model (QSMC_NEP doubles, at least one): pass exactly those bytes
    UserEpArg epa;
    memset(&epa, 0, sizeof(epa));
    for (int k = 0; k < uk->n_ep; ++k) epa.v[k] = ep[k];
    long long ldx_ = ldx, n_ = n, outcome_ = outcome;
    double *partials = h->partials;
    void *args[] = {(void *)&x, (void *)&ldx_, (void *)&n_, (void *)&w_in, (void *)&w_out, (void *)&prev_norm, (void *)&epa,
                    (void *)&outcome_, (void *)&partials, (void *)&vec};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    prof_events(h, w_in ? QSMC_PROF_UPDATE : QSMC_PROF_UPDATE_ONES, &e0, &e1);
    if (e0) HIP_TRY(h, hipEventRecord(e0, s));
    HIP_TRY(h, hipModuleLaunchKernel(uk->upd, (unsigned)grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    if (e1) HIP_TRY(h, hipEventRecord(e1, s));
    rc = launch_reduce(h, ns, grid, ro, s);
    if (rc) return rc;
    return collect_stats(h, ns, stats_host, moments_host, n_mom, s);
}

// qsmc_update_multi's contract for a compiled user model: k <= 8 data of one batch_update window in one pass
int qsmc_update_multi_user(qsmc_handle_t h, qsmc_user_kernel_t uk, const double *x, int64_t ldx, int64_t n, const double *w_in,
                           double *w_out, double prev_norm, const double *eps, const int64_t *outcomes, int32_t k,
                           qsmc_update_stats_t *stats_host, double *moments_host, qsmc_stream_t stream) {
    if (h) { h->prep.valid = 0; h->rsq.valid = 0; h->canon_next.kind = 0; h->expect_next = 0.0; h->ts.w = nullptr; }
    if (!h || !uk || !x || !w_out || !outcomes || !stats_host || n <= 0 || k < 1 || k > MULTI_KMAX || (uk->n_ep > 0 && !eps))
        return QSMC_ERR_INVALID;
    const int d = uk->d, dmom = d <= 4 ? d : 0;
    if (moments_host && !dmom) return QSMC_ERR_UNSUPPORTED;
    const int n_mom = dmom + dmom * (dmom + 1) / 2, ns = 3 * MULTI_KMAX + n_mom;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n, 256 * 8);
    int rc = ensure_partials(h, (size_t)grid * (ns + 1));
    if (rc) return rc;
    ReduceOut ro = make_reduce(h, true, nullptr);
    ++h->ts.gen;
    h->ts.armed = 0;
    h->spec.launched = 0;
    // QsmcUserWindow of kernels/user_jit.hpp: ep[8][max(n_ep, 1)], outcome[8], k -- built to that layout here
    const int nep1 = uk->n_ep > 0 ? uk->n_ep : 1;
    unsigned char win[8 * USER_MAX_EP * sizeof(double) + 8 * sizeof(long long) + 16];
    memset(win, 0, sizeof(win));
    double *wep = reinterpret_cast<double *>(win);
    for (int j = 0; j < k; ++j)
        for (int q = 0; q < uk->n_ep; ++q) wep[(size_t)j * nep1 + q] = eps[(size_t)j * uk->n_ep + q];
    long long *woc = reinterpret_cast<long long *>(win + (size_t)8 * nep1 * sizeof(double));
    for (int j = 0; j < k; ++j) woc[j] = outcomes[j];
    int *wk = reinterpret_cast<int *>(win + (size_t)8 * nep1 * sizeof(double) + 8 * sizeof(long long));
    *wk = k;
    long long ldx_ = ldx, n_ = n;
    double *partials = h->partials;
    void *args[] = {(void *)&x, (void *)&ldx_, (void *)&n_, (void *)&w_in, (void *)&w_out, (void *)&prev_norm, (void *)win,
                    (void *)&partials};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    prof_events(h, QSMC_PROF_UPDATE_MULTI, &e0, &e1);
    if (e0) HIP_TRY(h, hipEventRecord(e0, s));
    HIP_TRY(h, hipModuleLaunchKernel(uk->upd_multi, (unsigned)grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    if (e1) HIP_TRY(h, hipEventRecord(e1, s));
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

// L_out[o][e][i] = likelihood(x_i; eps[e], outcomes[o]): qsmc_likelihood's layout for a compiled user model
int qsmc_likelihood_user(qsmc_handle_t h, qsmc_user_kernel_t uk, const double *x, int64_t ldx, int64_t n, const double *eps,
                         int32_t n_e, const int64_t *outcomes, int32_t n_o, double *L_out, qsmc_stream_t stream) {
    if (!h || !uk || !x || !outcomes || !L_out || n <= 0 || n_e < 1 || n_o < 1 || (uk->n_ep > 0 && !eps)) return QSMC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n, 256 * 4);
    long long ldx_ = ldx, n_ = n;
    for (int o = 0; o < n_o; ++o)
        for (int e = 0; e < n_e; ++e) {
            UserEpArg epa;
            memset(&epa, 0, sizeof(epa));
            for (int k = 0; k < uk->n_ep; ++k) epa.v[k] = eps[(size_t)e * uk->n_ep + k];
            long long outcome_ = outcomes[o];
            double *dst = L_out + ((size_t)o * n_e + e) * (size_t)n;
            void *args[] = {(void *)&x, (void *)&ldx_, (void *)&n_, (void *)&epa, (void *)&outcome_, (void *)&dst};
            HIP_TRY(h, hipModuleLaunchKernel(uk->lik, (unsigned)grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
        }
    return QSMC_OK;
}

int qsmc_valid_user(qsmc_handle_t h, qsmc_user_kernel_t uk, const double *x, int64_t ldx, int64_t n, uint8_t *mask_out,
                    qsmc_stream_t stream) {
    if (!h || !uk || !x || !mask_out || n <= 0) return QSMC_ERR_INVALID;
    const int grid = grid_for(n, 256 * 4);
    long long ldx_ = ldx, n_ = n;
    void *args[] = {(void *)&x, (void *)&ldx_, (void *)&n_, (void *)&mask_out};
    HIP_TRY(h, hipModuleLaunchKernel(uk->valid, (unsigned)grid, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr));
    return QSMC_OK;
}

// ---- host: the shard plan of a sharded resample ------------------------------------------------------------
// T ~ Multinomial(n_total; W_h / sum W): how many of the n_total children descend from shard h.  Every rank draws it
// from the same (seed, epoch) and gets the same answer, so planning a resample needs no communication (the W_h came
// with the update's sums).  G - 1 conditional binomials, T_h | T_<h ~ Binomial(n_left, W_h / W_>=h); the binomial is
// exact: sequential inversion while n min(p, 1 - p) < 30, else the BTPE rejection algorithm (Kachitvichyanukul &
// Schmeiser, "Binomial random variate generation", CACM 31 (1988) 216-222, steps 0-6: triangle / parallelogram /
// two exponential tails as the majorising function, squeeze, explicit recursion near the mode, Stirling bound far
// from it).  Uniforms: Philox4x32-10, key = seed, counter (draw index, epoch, stream 0x504C414E "PLAN"): two per block.
struct PlanRng {
    uint32_t k0, k1, epoch_lo, epoch_hi;
    uint64_t block;
    int have;
    double spare;
    double next() {
        if (have) { have = 0; return spare; }
        const U4 r = philox4x32_10(U4{(uint32_t)block, (uint32_t)(block >> 32) ^ epoch_hi, epoch_lo, 0x504C414Eu}, k0, k1);
        ++block;
        spare = u53(r.z, r.w);
        have = 1;
        return u53(r.x, r.y);
    }
};

static int64_t binomial_inversion(PlanRng &g, int64_t n, double p) {     // n p < 30, p <= 1/2
    const double q = 1.0 - p, s = p / q, a = (double)(n + 1) * s;
    const double r0 = exp((double)n * log(q));                            // Pr(0); n p < 30 keeps it above e^-60
    for (;;) {
        double r = r0, u = g.next();
        int64_t x = 0;
        bool ok = true;
        while (u > r) {
            u -= r;
            ++x;
            if (x > n || x > 4096) { ok = false; break; }                 // (rounding at the far tail: draw again)
            r *= a / (double)x - s;
        }
        if (ok) return x;
    }
}

static double btpe_stirling(double t2) {                                  // the correction series of step 5.3
    return (13860.0 - (462.0 - (132.0 - (99.0 - 140.0 / t2) / t2) / t2) / t2) / 166320.0;
}

static int64_t binomial_btpe(PlanRng &g, int64_t n, double p) {           // n p >= 30, p <= 1/2
    const double r = p, q = 1.0 - r, nd = (double)n, nrq = nd * r * q;
    const double fm = nd * r + r;
    const int64_t M = (int64_t)floor(fm);
    const double p1 = floor(2.195 * sqrt(nrq) - 4.6 * q) + 0.5;
    const double xm = (double)M + 0.5, xl = xm - p1, xr = xm + p1;
    const double c = 0.134 + 20.5 / (15.3 + (double)M);
    double al = (fm - xl) / (fm - xl * r);
    const double laml = al * (1.0 + 0.5 * al);
    al = (xr - fm) / (xr * q);
    const double lamr = al * (1.0 + 0.5 * al);
    const double p2 = p1 * (1.0 + 2.0 * c), p3 = p2 + c / laml, p4 = p3 + c / lamr;
    for (;;) {
        const double u = g.next() * p4;
        double v = g.next();
        int64_t y;
        if (u <= p1) {                                                    // 1: the triangle -- accept at once
            return (int64_t)floor(xm - p1 * v + u);
        } else if (u <= p2) {                                             // 2: the parallelograms
            const double x = xl + (u - p1) / c;
            v = v * c + 1.0 - fabs((double)M - x + 0.5) / p1;
            if (v > 1.0 || v <= 0.0) continue;
            y = (int64_t)floor(x);
        } else if (u <= p3) {                                             // 3: left exponential tail
            if (v <= 0.0) continue;
            y = (int64_t)floor(xl + log(v) / laml);
            if (y < 0) continue;
            v = v * (u - p2) * laml;
        } else {                                                          // 4: right exponential tail
            if (v <= 0.0) continue;
            y = (int64_t)floor(xr - log(v) / lamr);
            if (y > n) continue;
            v = v * (u - p3) * lamr;
        }
        // 5: accept y with probability f(y) / (majorising function), f(y) = Pr(y) / Pr(M)
        const int64_t k = y > M ? y - M : M - y;
        if (k <= 20 || (double)k >= 0.5 * nrq - 1.0) {                    // 5.1: f(y) by the recursion from the mode
            const double s = r / q, a = s * (nd + 1.0);
            double F = 1.0;
            if (M < y) for (int64_t i = M + 1; i <= y; ++i) F *= a / (double)i - s;
            else if (M > y) for (int64_t i = y + 1; i <= M; ++i) F /= a / (double)i - s;
            if (v > F) continue;
            return y;
        }
        const double kd = (double)k;                                      // 5.2: squeeze on ln f(y)
        const double rho = (kd / nrq) * ((kd * (kd / 3.0 + 0.625) + 0.1666666666666) / nrq + 0.5);
        const double t = -kd * kd / (2.0 * nrq), A = log(v);
        if (A < t - rho) return y;
        if (A > t + rho) continue;
        const double x1 = (double)(y + 1), f1 = (double)(M + 1), z = nd + 1.0 - (double)M, w = nd - (double)y + 1.0;   // 5.3
        const double bound = xm * log(f1 / x1) + (nd - (double)M + 0.5) * log(z / w) + (double)(y - M) * log(w * r / (x1 * q)) +
                             btpe_stirling(f1 * f1) / f1 + btpe_stirling(z * z) / z + btpe_stirling(x1 * x1) / x1 +
                             btpe_stirling(w * w) / w;
        if (A > bound) continue;
        return y;
    }
}

static int64_t binomial_exact(PlanRng &g, int64_t n, double p) {
    if (n <= 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    const bool flip = p > 0.5;
    const double r = flip ? 1.0 - p : p;
    const int64_t y = (double)n * r < 30.0 ? binomial_inversion(g, n, r) : binomial_btpe(g, n, r);
    return flip ? n - y : y;
}

int qsmc_shard_plan_totals(uint64_t seed, uint64_t epoch, const double *shard_weights, int32_t n_shards, int64_t n_total,
                           int64_t *totals_out) {
    if (!shard_weights || !totals_out || n_shards < 1 || n_total < 0) return QSMC_ERR_INVALID;
    double rest = 0.0;
    for (int h = 0; h < n_shards; ++h) {
        if (!(shard_weights[h] >= 0.0) || !std::isfinite(shard_weights[h])) return QSMC_ERR_INVALID;
        rest += shard_weights[h];
    }
    if (!(rest > 0.0)) return QSMC_ERR_INVALID;
    PlanRng g{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)epoch, (uint32_t)(epoch >> 32), 0ull, 0, 0.0};
    int64_t left = n_total;
    for (int h = 0; h < n_shards; ++h) {
        // tail mass summed from the back would be more accurate; with <= 64 non-negative terms the running difference is
        // good to a few ulp, and the last shard with weight takes what is left so that the total is exact
        double tail = 0.0;
        for (int j = h + 1; j < n_shards; ++j) tail += shard_weights[j];
        int64_t t;
        if (left == 0 || shard_weights[h] == 0.0) t = 0;
        else if (tail == 0.0) t = left;
        else t = binomial_exact(g, left, shard_weights[h] / (shard_weights[h] + tail));
        totals_out[h] = t;
        left -= t;
    }
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
                return QSMC_ERR_TIMEOUT;                       // a peer did not arrive
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
    // kernels/sqrtm.hpp: the routine the device wavefront mirrors (work arrays on the stack for d <= 16, heap above)
    return sqrtm_psd_host(A, d, scale, S_out, err_out) ? QSMC_OK : QSMC_ERR_ALLOC;
}

}  // extern "C"
