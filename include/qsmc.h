/* qsmc.h -- C ABI of libqsmc_hip.so: the MI355X (gfx950) sequential-Monte-Carlo hot path.
 *
 * QInfer (the reference) is pure Python and has no FFI; the boundary this library sits behind is
 * the Python class surface of SMCUpdater / LiuWestResampler / Model.
 * Each entry point below names the reference code it replaces (paths relative to
 * /root/reference/src/qinfer/).  INTEGRATION.md shows the ctypes stub a QInfer maintainer
 * would add to bind them.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (qsmc_strerror); nothing throws;
 *  - all particle buffers are CALLER-OWNED DEVICE memory, float64;
 *  - particle locations are SoA: x[m * ldx + i] is parameter m of particle i (0 <= m < d);
 *  - `stream` is a hipStream_t passed as void*; calls are asynchronous on it unless they return
 *    host values (documented per function), in which case they synchronise that stream;
 *  - weights are kept UNNORMALISED on the device: the true weight of particle i is
 *    w[i] / norm, where norm is the running normaliser (sum of w) the caller carries as a host
 *    double; this is QInfer's `hyp_weights / norm_scale` (smc.py:354-373) with the division
 *    deferred to the next read, so no extra pass over HBM is spent on renormalising;
 *  - no hidden global state: scratch lives in the opaque handle (one handle per device/stream).
 */
#ifndef QSMC_H
#define QSMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QSMC_ABI_VERSION 3
#define QSMC_MAX_D 16            /* largest n_modelparams the narrow kernels take (2-qubit tomography): a particle, the mean and
                                    S ride in registers / the kernarg segment */
#define QSMC_MAX_D_WIDE 64       /* largest n_modelparams with native kernels at all: QSMC_MODEL_TOMOGRAPHY with 16 < d <= 64
                                    (dim 5 .. 8; three qubits: d = 64) takes the wide kernels (csrc/kernels/wide.hpp) in
                                    qsmc_likelihood, qsmc_update_fused, qsmc_step (update and tests; a due resample is the
                                    caller's), qsmc_moments, qsmc_lw_centres / _perturb, qsmc_lw_resample_philox (+ _prepare)
                                    and qsmc_tomo_canonicalize2; qsmc_update_multi takes windows whose measurement vectors have at
                                    most four nonzero entries each (a Pauli measurement has two) and answers
                                    QSMC_ERR_UNSUPPORTED otherwise, as the qsmc_hypothetical_sums_* family does above
                                    QSMC_MAX_D (callers loop qsmc_update_fused / use qsmc_likelihood) */

typedef struct qsmc_ctx *qsmc_handle_t;
typedef void *qsmc_stream_t;     /* hipStream_t */

enum qsmc_status {
    QSMC_OK = 0,
    QSMC_ERR_INVALID = -1,       /* bad argument (null pointer, d out of range, unknown model kind) */
    QSMC_ERR_HIP = -2,           /* a HIP runtime call failed; see qsmc_last_hip_error */
    QSMC_ERR_ALLOC = -3,
    QSMC_ERR_UNSUPPORTED = -4,
    QSMC_ERR_TIMEOUT = -5        /* a peer of a host collective (qsmc_host_allgather / _allreduce) did not arrive in time */
};

/* Model kinds with a native likelihood kernel. */
enum qsmc_model_kind {
    QSMC_MODEL_PRECESSION = 1,          /* test_models.py:64-213  SimpleInversion/SimplePrecession  */
    QSMC_MODEL_BINOMIAL_PRECESSION = 2, /* derived_models.py:222-360 BinomialModel(SimplePrecession) */
    QSMC_MODEL_RB = 3,                  /* rb.py:81-195 RandomizedBenchmarkingModel()                */
    QSMC_MODEL_RB_INTERLEAVED = 4,      /* rb.py:81-195 (interleaved=True)                           */
    QSMC_MODEL_TOMOGRAPHY = 5,          /* tomography/models.py:82-226 TomographyModel               */
    QSMC_MODEL_BINOMIAL_RB = 6,         /* BinomialModel(RandomizedBenchmarkingModel()): simple_est.py:212 */
    QSMC_MODEL_BINOMIAL_RB_INTERLEAVED = 7, /* ... (interleaved=True)                                 */
    QSMC_MODEL_UNKNOWN_T2 = 8           /* test_models.py:222-259 UnknownT2Model (omega, 1/T2)        */
};

typedef struct qsmc_model {
    int32_t kind;                /* enum qsmc_model_kind */
    int32_t d;                   /* n_modelparams */
    double  min_freq;            /* precession validity: omega > min_freq (test_models.py:109-110) */
    int32_t postselect_all_valid;/* 1: are_models_valid == True everywhere (tomography :143-147) */
    int32_t reserved;
    double  likelihood_power;    /* MLEModel(model, gamma): L ** gamma (derived_models.py:673-691); 0 = plain */
} qsmc_model_t;

/* One experiment (one row of the model's `expparams` record array). */
typedef struct qsmc_expparam {
    double   t;                  /* precession: evolution time (expparams['t'] / scalar)          */
    double   w_;                 /* inversion model reference frequency (0 for SimplePrecession)   */
    uint64_t n_meas;             /* binomial: expparams['n_meas']                                  */
    uint64_t m;                  /* RB: sequence length expparams['m']                             */
    int32_t  reference;          /* RB interleaved: expparams['reference']                         */
    int32_t  reserved;
    double   meas[QSMC_MAX_D];   /* tomography: expparams['meas'] (length d), d <= QSMC_MAX_D       */
    const double *meas_wide;     /* tomography, d > QSMC_MAX_D: expparams['meas'] (length d), HOST memory read during
                                    the call; meas[] is ignored then.  NULL otherwise.               */
} qsmc_expparam_t;

/* Per-update reduction results (all over the UNNORMALISED new weights w' = (w/norm) * L). */
typedef struct qsmc_update_stats {
    double sum;                  /* sum_i w'_i          == norm_scale of smc.py:357                */
    double sumsq;                /* sum_i w'_i^2        -> n_ess = sum^2 / sumsq (distributions.py:307) */
    double min;                  /* min_i w'_i          -> negative-weight guard smc.py:416-418    */
    double n_bad;                /* #{i : !(w'_i >= 0)} (counts NaN too, like np.all(w >= 0))      */
} qsmc_update_stats_t;

/* ---- library ---------------------------------------------------------------------------- */
int         qsmc_abi_version(void);
const char *qsmc_strerror(int status);
const char *qsmc_last_hip_error(qsmc_handle_t h);
/* A handle owns scratch, the pinned completion word and the arrival tickets of its reducing kernels: it is bound to ONE
 * stream at a time.  Calls on a handle must come from one host thread at a time and their `stream` arguments must be
 * the same stream until that stream has been synchronised (launches of one stream are serialised; two streams sharing a
 * handle would interleave ticket arrivals and scratch use).  Use one handle per stream. */
int         qsmc_create(qsmc_handle_t *out, int device);
int         qsmc_destroy(qsmc_handle_t h);
/* Test hooks (process-wide, all off by default): each selects an independent form of a kernel that the parity tests
 * compare the default form against, or makes a rare branch common; none changes a result's law.
 *   QSMC_HOOK_MULTI_GENERIC    value != 0: k_update_multi sends every tile through its general path
 *   QSMC_HOOK_REDRAW_NO_SMALL  value != 0: the redraw kernel takes the global-CDF form also for short queues
 *   QSMC_HOOK_HYP_NO_CHAIN     value != 0: design passes of binomial experiments by the thread-per-particle kernel
 *   QSMC_HOOK_TOMO_DENSE       value != 0: tomography updates read all d rows also for sparse measurement vectors
 *   QSMC_HOOK_POISSON_MARGIN   value = kappa in lambda = n_out - kappa sqrt(n_out) of the bucketed counts (default 5)
 *   QSMC_HOOK_CANON_WIDE_JACOBI value != 0: canonicalize of dim 5 and 8 takes the one-lane eigenvector form for its listed
 *                              particles (k_tomo_canon_list_wide), not the one-sided form without eigenvectors (k_tomo_canon_list_os)
 * Returns QSMC_ERR_INVALID for an unknown hook. */
#define QSMC_HOOK_MULTI_GENERIC 1
#define QSMC_HOOK_REDRAW_NO_SMALL 2
#define QSMC_HOOK_HYP_NO_CHAIN 3
#define QSMC_HOOK_TOMO_DENSE 4
#define QSMC_HOOK_POISSON_MARGIN 5
#define QSMC_HOOK_CANON_WIDE_JACOBI 6
int         qsmc_test_hook(int32_t hook, double value);
/* Compute units this process can actually run on (a census taken by qsmc_create: under HSA_CU_MASK or a partitioned
 * part fewer than the device attribute reports) and the reported number.  The resampler's grid-barrier kernels
 * (k_bucket_counts, k_bucket_redraw) size their resident grids by the first. */
int         qsmc_device_cus(qsmc_handle_t h, int32_t *usable_out, int32_t *reported_out);

/* Kernel timing for bench.py's roofline line: when enabled, qsmc_update_fused brackets its main
 * kernel (not the finalize/copy) with hipEvents on `stream` (hipExtLaunchKernelGGL start/stop events:
 * the kernel's own execution, what rocprofv3 --kernel-trace reports).  Up to 4096 launches are kept
 * in a ring (the bucketed sampler's main kernel is timed the same way, tagged QSMC_PROF_SAMPLE);
 * qsmc_profile_read hands back their durations in milliseconds and tags, oldest first, and clears
 * the ring -- one call after the timed region, nothing per step.  qsmc_last_update_kernel_ms waits
 * for and returns the most recent one.
 * `enabled` = N > 1 times only every N-th launch of each tag: a launch that carries start/stop events
 * drains the queue around itself (~10 us per step at N = 1 in bench.py), a sampled average does not. */
int         qsmc_set_profiling(qsmc_handle_t h, int enabled);
/* Which tags (bit t = QSMC_PROF_* tag t) are timed while profiling is on; default: all.  bench.py's timed pass times the
 * dominant kernel only (the update: tags 0 and 2) -- every other event pair there is queue drain that the roofline
 * object does not need; the other kernels' durations come from the census pass right after it. */
int         qsmc_set_profiling_tags(qsmc_handle_t h, uint32_t tag_mask);
int         qsmc_last_update_kernel_ms(qsmc_handle_t h, float *ms_out);
int         qsmc_profile_read(qsmc_handle_t h, float *ms_out, int32_t *tags_out, int32_t cap, int32_t *n_out);
#define QSMC_PROF_UPDATE 0      /* k_update_fused, explicit weights (24 B/particle at d = 1) */
#define QSMC_PROF_SAMPLE 1      /* k_bucket_sample (the resampler's dominant kernel) */
#define QSMC_PROF_UPDATE_ONES 2 /* k_update_fused, implicit all-ones weights (16 B/particle at d = 1) */
#define QSMC_PROF_CANON_CLASSIFY 3 /* tomography canonicalize, pass 1 (k_tomo_classify; or the single-pass kernel) */
#define QSMC_PROF_CANON_LIST 4     /* tomography canonicalize, pass 2 (k_tomo_canon_list) */
#define QSMC_PROF_MOMENTS 5        /* weighted moments, 4 < d <= 16 (k_moments_mfma) */
#define QSMC_PROF_COUNTS 6         /* the resampler's chunk counts + plan launch (k_bucket_counts) */
#define QSMC_PROF_COUNTS_SKIPPED 7 /* a speculative k_bucket_counts that left at its gate (qsmc_lw_arm_prefix) */
#define QSMC_PROF_ANCESTORS 8      /* d = 16 sampler, first half: ancestors of every slot (k_bucket_anc16); tag 1 is its kick kernel */
#define QSMC_PROF_UPDATE_MULTI 10   /* k_update_multi: up to 8 data in one pass (batch_update's fused windows) */
#define QSMC_PROF_HYP_SUMS 11       /* k_hyp_sums: one hypothetical experiment, all outcomes (bayes_risk / expected_information_gain) */
#define QSMC_PROF_CANON_BUILD 12    /* canonicalize of dim 5 .. 8: rho_packed = Mb x on the matrix cores (k_gemm_wide<NB, 0>); tags 3 / 4 are its
                                       pivot test (k_tomo_ldl_wide) and its Jacobi pass (k_tomo_jacobi_wide) */
#define QSMC_PROF_CANON_EXPAND 13   /* ... and x = Me R_packed with the trace renormalisation (k_gemm_wide<NB, 1>) */
#define QSMC_PROF_NTAGS 16

/* ---- likelihood, contract form (abstract_model.py:444-468 + :666-686; a6-a10) ------------ */
/* L_out[(o * n_e + e) * n + i] = Pr(outcomes[o] | x_i ; exps[e]).  This is the
 * (outcomes, experiments, particles) layout smc.py:353 transposes to.  `exps`/`outcomes` are HOST. */
int qsmc_likelihood(qsmc_handle_t h, const qsmc_model_t *model,
                    const double *x, int64_t ldx, int64_t n,
                    const qsmc_expparam_t *exps, int32_t n_e,
                    const int64_t *outcomes, int32_t n_o,
                    double *L_out, qsmc_stream_t stream);

/* Model.are_models_valid (test_models.py:109-110, rb.py:149-176, tomography/models.py:143-147). */
int qsmc_are_models_valid(qsmc_handle_t h, const qsmc_model_t *model,
                          const double *x, int64_t ldx, int64_t n,
                          uint8_t *valid_out, qsmc_stream_t stream);

/* ---- fused Bayes update (smc.py:388-457 = hypothetical_update :324-386 + n_ess; a1-a3) --- */
/* w_out[i] = (w_in[i] / prev_norm) * Pr(outcome | x_i ; exp);  stats reduced in the same pass
 * (per-workgroup partials, then a one-workgroup kernel sums them in index order: deterministic).
 * w_out may alias w_in.  w_in == NULL stands for all-ones weights (uniform cloud after a resample or
 * reset, with prev_norm = N): the fill pass and 8 B/particle of reads are skipped, same arithmetic.
 * stats_dev (device, non-NULL to use): 4 doubles in qsmc_update_stats_t order followed, for d <= 4,
 * by the d + d(d+1)/2 moment sums described below (so a sharded caller can all-gather one device
 * vector per datum without a host round trip).  If stats_host or moments_host is non-NULL the call synchronises `stream`.
 * moments_host (d <= 4 only): [sum w' x_m (d), sum w' x_m x_n for m <= n row-major (d(d+1)/2)] of the
 * NEW unnormalised weights -- divide by stats.sum to get E[x], E[x x^T] (distributions.py:337-399)
 * without another pass over HBM. */
int qsmc_update_fused(qsmc_handle_t h, const qsmc_model_t *model,
                      const double *x, int64_t ldx, int64_t n,
                      const double *w_in, double *w_out, double prev_norm,
                      const qsmc_expparam_t *exp, int64_t outcome,
                      double *stats_dev, qsmc_update_stats_t *stats_host, double *moments_host,
                      qsmc_stream_t stream);

/* ---- one datum in ONE call (SMCUpdater.update, smc.py:388-457, with its resample test smc.py:263-277) -------
 * qsmc_update_fused + the part of `update` that follows it when no guard fires, in C, so that between two data the
 * host runs a handful of Python statements instead of a page of them (round 2: 13 us of Python per datum, the GPU
 * idle a quarter of the run):
 *   fused update of w_alt <- (w / norm) * L(x; exp, outcome) with its sums (and the moment sums, d <= 4);
 *   norm_scale = sum w' with the |.| < eps -> 1 fix (smc.py:357-370); negative / NaN weights (smc.py:416-418) or
 *   all-zero weights (smc.py:423-436) -> QSMC_STEP_GUARD: NOTHING is committed, the raw sums are in the struct and the
 *   caller runs its guard / policy code exactly as after qsmc_update_fused (w_alt holds the new weights);
 *   otherwise commit: w <-> w_alt swapped IN THE STRUCT, norm, sumsq, n_ess = norm^2 / sumsq (distributions.py:299-307),
 *   min_n_ess (smc.py:452-453); then, if check_for_resample: n_ess <= 10 -> QSMC_STEP_SMALL_ESS (the caller warns,
 *   smc.py:265-270); n_ess < ess_below -> QSMC_STEP_RESAMPLE_DUE.
 * A due resample is also QUEUED here when the caller allows it (lw.enabled, device-RNG Liu-West, d <= 4, whose
 * moments came with the update): mean = S1 / norm, cov = S2 / norm - mean mean^T (distributions.py:337-399), the
 * zero-covariance substitute (resamplers.py:283-290), S = h sqrtm_psd(cov) (utils.py:593-607), and
 * qsmc_lw_resample_philox into lw.x_out -- the very call the caller is about to make, a host round trip earlier
 * (QSMC_STEP_RESAMPLE_QUEUED).  The caller still forms mean / covariance / square root itself, with its warnings and
 * errors (resamplers.py:283-299), and calls qsmc_lw_resample_philox as always: called with bit-identical arguments
 * (and x_out = lw.x_out) it finds its work done and returns at once; with anything different the resample is simply
 * run again into the caller's buffer.  The queued resample never replaces the caller's decision, it only starts it
 * early.
 * d = 16 (2-qubit tomography; moments are a pass of their own, k_moments_mfma): the call queues the moments and the
 * first half of the split sampler (k_bucket_anc16: ancestors, needs weights only), waits for the moments, forms mean /
 * covariance / S on the host while that kernel runs, and queues the second half (k_bucket_kick16) behind it; the moments
 * come back in moments_big (qsmc_moments' out_host layout) so the caller need not compute them again.  In round 2 the
 * GPU idled ~120 us per d = 16 resample while Python did this between two launches.
 * The speculative weight-only prefix of qsmc_lw_arm_prefix is armed from lw.* by every call with lw.prefix != 0.
 * w == NULL: implicit all-ones weights (then w_alt becomes NULL on commit: supply a buffer before the next call). */
#define QSMC_STEP_GUARD           1
#define QSMC_STEP_SMALL_ESS       2
#define QSMC_STEP_RESAMPLE_DUE    4
#define QSMC_STEP_RESAMPLE_QUEUED 8
#define QSMC_STEP_PLAN_READY      16
#define QSMC_STEP_PREFIX_QUEUED   32
#define QSMC_STEP_MAX_RANKS 64
typedef struct qsmc_step_lw {
    int32_t  enabled;            /* queue the resample when it is due (needs x_out)                         */
    int32_t  prefix;             /* arm the gated weight-only prefix behind every update (qsmc_lw_arm_prefix) */
    int32_t  postselect, maxiter;
    double   a, h, zero_cov_comp;
    uint64_t seed, epoch;        /* of the resample that would follow                                       */
    int64_t  n_out;
    double  *x_out;              /* device, SoA d x n_out, row stride ldx_out                               */
    int64_t  ldx_out;
    int32_t  canon_kind;         /* d = 16 tomography, canonicalize folded into the resample (qsmc_lw_fuse_canonicalize): */
    int32_t  canon_allow_sub;    /*   0 = no, 1 = the 2-qubit Pauli basis, 2 = a dense basis; allow_subnormalized         */
    const double *canon_basis;   /*   device basis tensor (dense) or NULL (Pauli)                                         */
    int64_t  redraws_seen;       /* first tries of this cloud's LAST resample that failed postselection (maintained by    */
    int32_t  redraw_pending;     /*   qsmc_step: read back with the next update's sums while redraw_pending != 0; the      */
                                 /*   caller sets redraw_pending after a resample of its own, 0 / 0 after a reset): what  */
                                 /*   the next resample is told to expect (qsmc_lw_expect_redraws)                         */
    int32_t  adopt;              /* != 0: a resample queued by this call IS the caller's (it will not repeat the call to  */
                                 /*   qsmc_lw_resample_philox; the flags it needs for the reference's warnings are in the */
                                 /*   struct: cov, cov_lambda_min, S_err): counted when the caller says qsmc_step_adopted  */
} qsmc_step_lw_t;
typedef struct qsmc_step {
    /* the cloud -- kept current by the caller; w / w_alt / norm / sumsq / min_n_ess advance here on commit */
    const double *x; int64_t ldx; int64_t n;
    const double *w;             /* unnormalised weights, true weight w / norm; NULL: all ones              */
    double       *w_alt;         /* where the new weights go                                                */
    double        norm, sumsq, min_n_ess;
    /* tests */
    double        zero_weight_thresh, ess_below;
    int32_t       check_for_resample, reserved0;
    qsmc_step_lw_t lw;
    /* results of the latest call */
    int32_t       status, reserved1;
    uint64_t      update_token;  /* qsmc_update_token after this update                                     */
    qsmc_update_stats_t stats;   /* raw sums of the new weights                                             */
    double        n_ess;
    double        moments[14];   /* d <= 4: [sum w' x_m, upper(sum w' x_m x_n)] of the new weights           */
    double        mean[QSMC_MAX_D], cov[QSMC_MAX_D * QSMC_MAX_D], S[QSMC_MAX_D * QSMC_MAX_D], S_err;   /* of a queued resample */
    double        moments_big[1 + QSMC_MAX_D + QSMC_MAX_D * (QSMC_MAX_D + 1) / 2];   /* d > 4, a queued resample: qsmc_moments' out_host */
    /* A SHARD of a cloud held by `ex_world` processes of one host (one per GPU): with ex_segment != NULL the step makes
     * the sharded updater's one per-datum collective itself -- qsmc_host_allreduce (below) of [sum w', sum w'^2, min,
     * #bad, moment sums] right after this shard's sums arrive -- and everything after it (guards, commit, n_ess, the
     * resample test: smc.py:369-457) runs on the GLOBAL sums, which is what `stats`, `moments`, `norm`, `sumsq`, `n_ess`
     * then hold; `n` stays this shard's size, `ess_below` / `min_n_ess` are the caller's global figures.  *ex_k is the
     * exchange's call counter (shared with the caller's other qsmc_host_* calls on the segment; advanced here).
     * shard_sums[r] = shard r's sum w' (the next resample plan's input).  lw.enabled / lw.prefix must be 0: a sharded
     * resample needs the shard plan, which is the caller's. */
    void         *ex_segment;
    int32_t       ex_rank, ex_world, ex_max_len, ex_reserved;
    uint64_t     *ex_k;
    double        ex_timeout_s;
    double        shard_sums[QSMC_STEP_MAX_RANKS];
    /* ... and, with plan_enabled, the first moves of a due resample: the shard plan (qsmc_shard_plan_totals with
     * plan_seed / plan_epoch over shard_sums[] and plan_n_total; status gets QSMC_STEP_PLAN_READY, plan_totals[] the plan)
     * and, when every shard's share is within plan_tol of the balanced size n_total / world and none is empty
     * (plan_stay = 1: children stay with their ancestors, parallel.py), the resampler's weight-only prefix for THIS shard
     * -- qsmc_lw_resample_prepare(w, n, shard_sums[rank], plan_totals[rank], plan_prefix_seed, plan_epoch) from this
     * update's tile sums -- queued at once (QSMC_STEP_PREFIX_QUEUED); the caller's resample finds both done. */
    int32_t       plan_enabled, plan_stay;
    uint64_t      plan_seed, plan_epoch, plan_prefix_seed;
    int64_t       plan_n_total;
    double        plan_tol;
    int64_t       plan_totals[QSMC_STEP_MAX_RANKS];
    /* of a queued resample: the smallest eigenvalue of the covariance the square root was formed from (the Jacobi's own
     * diagonal) -- the caller's positive-semidefiniteness check (distributions.py:392-399: la.eig(cov) >= 0) without a
     * second eigendecomposition on the host while the GPU waits */
    double        cov_lambda_min;
} qsmc_step_t;
int qsmc_step(qsmc_handle_t h, qsmc_step_t *st, const qsmc_model_t *model, const qsmc_expparam_t *exp,
              int64_t outcome, qsmc_stream_t stream);
/* how many resamples qsmc_step queued on this handle, and how many of them the caller's own call adopted */
int qsmc_step_stats(qsmc_handle_t h, int64_t *n_queued, int64_t *n_adopted);
/* lw.adopt callers: "the resample the latest qsmc_step queued is mine" (SMCUpdater._adopt_queued) -- counted as adopted
 * HERE, by the party that decides; a caller that declines (its resampler's a / h / seed / epoch were edited in place
 * after the struct was filled) simply runs its own resample and the queued one shows up as queued, not adopted. */
int qsmc_step_adopted(qsmc_handle_t h);

/* batch_update fast path (smc.py:459-487): k <= 8 data applied in ONE pass over the cloud,
 * w_out[i] = (w_in[i] / prev_norm) * prod_j Pr(outcomes[j] | x_i ; exps[j]).
 * stats_host[j] holds the cumulative sums after datum j: sum = S_j, sumsq = Q_j, n_bad = #{!(w >= 0)}
 * at that datum (min is the minimum over the whole window), from which the caller forms
 * normalization_record[j] = S_j / S_{j-1} and n_ess_j = S_j^2 / Q_j exactly as the per-datum loop
 * would.  moments_host as in qsmc_update_fused (of the final weights).  Synchronises.  w_out must
 * not alias w_in if the caller wants to be able to discard the window (guards tripped). */
int qsmc_update_multi(qsmc_handle_t h, const qsmc_model_t *model,
                      const double *x, int64_t ldx, int64_t n,
                      const double *w_in, double *w_out, double prev_norm,
                      const qsmc_expparam_t *exps, const int64_t *outcomes, int32_t k,
                      qsmc_update_stats_t *stats_host, double *moments_host, qsmc_stream_t stream);

/* Experiment design (smc.py:553-663 bayes_risk / expected_information_gain): for ONE hypothetical
 * experiment and n_o outcomes, one pass over the cloud, nothing materialised.  With w~ = w / norm
 * (w == NULL: all-ones) and c = shift (pass the current mean; removes one-pass cancellation):
 *   out_host[o][0]         = sum w~ L_o                       hypothetical normalisation N[o]
 *   out_host[o][1]         = sum w~ L_o log L_o  (0 log 0 := 0)        N KLD = [1] - [0] log [0]
 *   out_host[o][2 + m]     = sum w~ L_o (x_m - c_m)           (d <= 4 only)
 *   out_host[o][2 + d + m] = sum w~ L_o (x_m - c_m)^2         N var = sum_m Q_m ([2+d+m] - [2+m]^2 / [0])
 * Row length is 2 + 2 d for d <= 4, else 2; a tomography model's rows have 2 columns whatever its d (its kernels are built
 * for the maximal dimension: a one-qubit model, d = 4, included).  Synchronises. */
int qsmc_hypothetical_sums(qsmc_handle_t h, const qsmc_model_t *model,
                           const double *x, int64_t ldx, int64_t n, const double *w, double norm,
                           const qsmc_expparam_t *exp, const int64_t *outcomes, int32_t n_o,
                           const double *shift, double *out_host, qsmc_stream_t stream);

/* The same for n_e experiments in one call, with the caller saying which columns it will read -- bayes_risk
 * (smc.py:553-611) uses [0] and the moment columns, expected_information_gain (smc.py:613-663) [0] and [1]: `what` =
 * QSMC_HYP_LOG | QSMC_HYP_MOMENTS bits; column [0] is always formed.  Experiment e has n_o[e] outcomes; `outcomes` holds
 * them one experiment after the other and out_host the rows in the same order (sum n_o rows of 2 + 2 d, or 2).  For
 * binomial experiments over consecutive outcomes the columns not asked for are not computed (no logarithm per particle
 * for the moments; no moments for the logarithm) and come back as NaN, and the experiments' passes are queued back to
 * back with ONE wait at the end; other models form every column whatever `what` says, an experiment at a time.
 * qsmc_hypothetical_sums is this call with n_e = 1 and both bits set.  Synchronises. */
#define QSMC_HYP_LOG 1
#define QSMC_HYP_MOMENTS 2
int qsmc_hypothetical_sums_multi(qsmc_handle_t h, const qsmc_model_t *model,
                                 const double *x, int64_t ldx, int64_t n, const double *w, double norm,
                                 const qsmc_expparam_t *exps, int32_t n_e, const int64_t *outcomes, const int32_t *n_o,
                                 const double *shift, int32_t what, double *out_host, qsmc_stream_t stream);

/* qsmc_hypothetical_sums_multi in two halves, so that a caller that prepares its experiments one after the other (a
 * design of n_e experiments: ~10 us of host work each) need not have prepared them all before the first pass starts:
 * _begin queues the passes of the experiments it is given -- same arguments, out_host rows must stay valid -- and returns
 * without waiting (experiments the two-ended walk does not serve are computed at once, as in _multi); any number of
 * _begin calls may follow one another; _collect waits for everything queued and fills the rows.  No other call on the
 * handle between the first _begin and _collect.  _multi is _begin + _collect. */
int qsmc_hypothetical_sums_begin(qsmc_handle_t h, const qsmc_model_t *model,
                                 const double *x, int64_t ldx, int64_t n, const double *w, double norm,
                                 const qsmc_expparam_t *exps, int32_t n_e, const int64_t *outcomes, const int32_t *n_o,
                                 const double *shift, int32_t what, double *out_host, qsmc_stream_t stream);
int qsmc_hypothetical_sums_collect(qsmc_handle_t h, qsmc_stream_t stream);

/* ---- user models compiled at run time ------------------------------------------------------------------------------------
 * The plugin contract Model.likelihood(outcomes, modelparams, expparams) (abstract_model.py:444-468) for a model the
 * library has no kernel for, at the speed of one that it has: the model states its per-particle likelihood as HIP device
 * source,
 *     __device__ double likelihood(const double *x, const double *ep, long long outcome);   // Pr(outcome | x; ep)
 *     __device__ bool valid(const double *x);     // optional (are_models_valid); announce: #define QSMC_USER_HAS_VALID 1
 * with x[0 .. d) one particle and ep[0 .. n_ep) the experiment's record fields as doubles in dtype order (QSMC_D and QSMC_NEP
 * are predefined).  qsmc_user_kernel_build compiles it with hiprtc INTO the fused update kernel for the handle's device
 * (python-qinfer_amd/csrc/kernels/user_jit.hpp; hiprtc_path: optional path of the libhiprtc.so to use when none is loaded
 * yet; log_out receives the compiler's log -- also on success, warnings).  d <= QSMC_MAX_D, n_ep <= 32.
 * QSMC_ERR_INVALID: the source does not compile; QSMC_ERR_UNSUPPORTED: no hiprtc on this machine. */
typedef struct qsmc_user_kernel *qsmc_user_kernel_t;
int qsmc_user_kernel_build(qsmc_handle_t h, const char *user_source, int32_t d, int32_t n_ep, const char *hiprtc_path,
                           qsmc_user_kernel_t *out, char *log_out, int32_t log_cap);
int qsmc_user_kernel_destroy(qsmc_user_kernel_t uk);
/* qsmc_update_fused for a compiled user model: one pass, w_out = (w_in / prev_norm) * likelihood(x; ep, outcome)
 * (w_in == NULL: all-ones weights), [sum, sum of squares, min, #bad] in stats_host and, for d <= 4, the packed weighted
 * moment sums in moments_host (same layout as qsmc_update_fused).  ep: n_ep doubles on the host. */
int qsmc_update_user(qsmc_handle_t h, qsmc_user_kernel_t uk, const double *x, int64_t ldx, int64_t n, const double *w_in,
                     double *w_out, double prev_norm, const double *ep, int64_t outcome, double *stats_dev,
                     qsmc_update_stats_t *stats_host, double *moments_host, qsmc_stream_t stream);
/* qsmc_update_multi for a compiled user model: the k <= 8 data of one batch_update window in one pass (eps: k rows of
 * n_ep doubles; stats_host[k]; moments_host as above, of the window's final weights). */
int qsmc_update_multi_user(qsmc_handle_t h, qsmc_user_kernel_t uk, const double *x, int64_t ldx, int64_t n, const double *w_in,
                           double *w_out, double prev_norm, const double *eps, const int64_t *outcomes, int32_t k,
                           qsmc_update_stats_t *stats_host, double *moments_host, qsmc_stream_t stream);
/* qsmc_likelihood for a compiled user model: L_out[n_o][n_e][n]; eps: n_e rows of n_ep doubles on the host. */
int qsmc_likelihood_user(qsmc_handle_t h, qsmc_user_kernel_t uk, const double *x, int64_t ldx, int64_t n, const double *eps,
                         int32_t n_e, const int64_t *outcomes, int32_t n_o, double *L_out, qsmc_stream_t stream);
/* qsmc_are_models_valid for a compiled user model (all ones if the source defines no valid()). */
int qsmc_valid_user(qsmc_handle_t h, qsmc_user_kernel_t uk, const double *x, int64_t ldx, int64_t n, uint8_t *mask_out,
                    qsmc_stream_t stream);

/* Same update for a model without a native kernel: L[i] was produced by the user's
 * Model.likelihood on the host and uploaded (plugin slow path). */
int qsmc_update_from_likelihood(qsmc_handle_t h, const double *L, int64_t n,
                                const double *w_in, double *w_out, double prev_norm,
                                double *stats_dev, qsmc_update_stats_t *stats_host,
                                qsmc_stream_t stream);

/* Negative-weight guard (smc.py:416-418): w[i] = clip(w[i] / norm, 0, 1), stats recomputed. */
int qsmc_clip_weights(qsmc_handle_t h, double *w, int64_t n, double norm,
                      double *stats_dev, qsmc_update_stats_t *stats_host, qsmc_stream_t stream);

/* Reduction only: stats of w / norm without writing (n_ess of an arbitrary cloud,
 * distributions.py:299-307). */
int qsmc_weight_stats(qsmc_handle_t h, const double *w, int64_t n, double norm,
                      double *stats_dev, qsmc_update_stats_t *stats_host, qsmc_stream_t stream);

/* The shard plan of a sharded resample (host only; no handle, no GPU): totals_out[h] = how many of the n_total new
 * particles descend from shard h, T ~ Multinomial(n_total; W_h / sum W), exact (conditional binomials: inversion /
 * BTPE on a Philox4x32-10 stream keyed by (seed, epoch)).  Every rank calls it with the same arguments -- the W_h
 * arrive with the update's sums (qsmc_host_allreduce rows / qsmc_step_t.shard_sums) -- and holds the same plan
 * without a collective.  Then shard h draws ITS totals_out[h] children from its local weights: given T the ancestors
 * are i.i.d. within a shard, which is resamplers.py:308-311's multinomial over the whole cloud, factorised. */
int qsmc_shard_plan_totals(uint64_t seed, uint64_t epoch, const double *shard_weights, int32_t n_shards,
                           int64_t n_total, int64_t *totals_out);

/* Host-side all-gather of n <= max_len doubles between the `world` processes of one host through a shared
 * memory segment (POSIX shm mapped by every rank; layout in qinfer_amd/parallel.py: HostExchange): call number k
 * (1, 2, ...; the same on every rank) writes vec into this rank's slot of bank k & 1, publishes k, spins until
 * every rank has published k, and copies the rank-ordered rows to rows_out[world][n].  No GPU involved: this
 * is the per-datum collective of the sharded updater (a handful of sums per rank), which is pure latency;
 * it stands where the reference gathers the whole likelihood array from its engines (parallel.py:216-224).
 * Returns QSMC_ERR_TIMEOUT if a peer has not arrived within timeout_s. */
int qsmc_host_allgather(void *segment, int32_t rank, int32_t world, int32_t max_len, uint64_t k, const double *vec,
                        int32_t n, double *rows_out, double timeout_s);

/* The same exchange with the reduction done here: tot_out[j] = sum over ranks, added in rank order (identical
 * bits on every rank), of entry j -- except entry min_index (>= 0), which is the minimum over ranks (NaN
 * propagates).  rows_out[world][n] still receives the rows (entry 0 of each row is that shard's weight total,
 * the input of the next resample plan).  The sharded SMCUpdater.update makes exactly one such call per datum
 * (smc.py:388-457: norm_scale, n_ess, the zero-weight guards; distributions.py:337-453: the moment sums). */
int qsmc_host_allreduce(void *segment, int32_t rank, int32_t world, int32_t max_len, uint64_t k, const double *vec,
                        int32_t n, int32_t min_index, double *rows_out, double *tot_out, double timeout_s);

/* ---- RCCL transport of the same reduction (SURVEY 8(b2) `qsmc_comm_init` / `qsmc_allreduce_sums`, 8(e)) -------
 * One process per GPU; the ranks of a sharded updater form an RCCL communicator inside the library.
 * qsmc_comm_unique_id: rank 0 fills id_out[128] (ncclGetUniqueId) and hands it to the others by any side channel
 * (the Python layer broadcasts it through torch.distributed); qsmc_comm_init: collective, every rank with the
 * same id; qsmc_comm_destroy: also done by qsmc_destroy.  librccl is bound at run time (dlsym), preferring the copy
 * already loaded in the process.
 * qsmc_allreduce_sums: vec_dev = this rank's n doubles on the DEVICE (the stats_dev vector qsmc_update_fused just
 * wrote: [sum w', sum w'^2, min w', #bad, moment sums...]); on `stream`, behind the kernel that produced it: ONE
 * ncclAllGather of the n entries of every rank, then a one-workgroup kernel that adds the rows IN RANK ORDER (entry
 * min_index, if >= 0: the minimum, NaN propagating) and publishes the totals; returns when they are in tot_host[n]
 * and firsts_host[nranks] (every rank's entry 0 = its shard's weight total, nullable).  n + nranks <= 188.  This is
 * the collective the reference's DirectViewParallelizedModel stands in for with a gather of the whole likelihood
 * array (parallel.py:216-224).  A gather moves bits without arithmetic and the summation order is fixed, so every
 * rank holds identical totals (and therefore takes the same n_ess / resample decision, smc.py:263-277), identical
 * also to qsmc_host_allreduce's on the same rows.
 * qsmc_comm_count: ranks in the communicator and this rank's index as RCCL itself reports them (ncclCommCount /
 * ncclCommUserRank) -- what bench.py records as `ranks_in_comm`. */
int qsmc_comm_unique_id(void *id_out);
int qsmc_comm_init(qsmc_handle_t h, int32_t rank, int32_t nranks, const void *unique_id);
int qsmc_comm_destroy(qsmc_handle_t h);
int qsmc_comm_count(qsmc_handle_t h, int32_t *nranks_out, int32_t *rank_out);
int qsmc_allreduce_sums(qsmc_handle_t h, const double *vec_dev, int32_t n, int32_t min_index, double *tot_host,
                        double *firsts_host, qsmc_stream_t stream);
/* What every rank does with the gathered rows, on rows the caller supplies (rows_dev: nranks vectors of n doubles, rank
 * after rank, device memory): the rank-ordered sums (entry min_index: the minimum), every rank's entry 0, published through
 * the handle's pinned block as qsmc_allreduce_sums publishes them.  No communicator involved: for a transport that
 * gathers by other means, and for testing the device half of the collective with any number of ranks on one GPU. */
int qsmc_publish_rows(qsmc_handle_t h, const double *rows_dev, int32_t n, int32_t min_index, int32_t nranks,
                      double *tot_host, double *firsts_host, qsmc_stream_t stream);

/* Sorting and searching for the posterior read-outs (est_credible_region, distributions.py:558-614;
 * posterior_marginal, smc.py:672-716).  qsmc_argsort: stable radix sort (rocPRIM) of n < 2^31 keys, ascending or
 * descending; keys_out and idx_out (the permutation, int64) are device arrays of n entries.
 * qsmc_searchsorted: out[k] = #{i : a[i] < q[k]} (side 0, NumPy 'left') or #{i : a[i] <= q[k]} (side 1, 'right')
 * for a non-decreasing device table a[0..n) and m device queries. */
int qsmc_argsort(qsmc_handle_t h, const double *keys, int64_t n, int32_t descending, double *keys_out,
                 int64_t *idx_out, qsmc_stream_t stream);
int qsmc_searchsorted(qsmc_handle_t h, const double *a, int64_t n, const double *q, int64_t m, int32_t side,
                      int64_t *out, qsmc_stream_t stream);

/* est_entropy (distributions.py:457-464): -sum over the particles with w_i / norm > 0 of (w_i / norm) log(w_i / norm),
 * one pass, result on the host.  w == NULL: implicit all-ones weights. */
int qsmc_weight_entropy(qsmc_handle_t h, const double *w, int64_t n, double norm, double *entropy_host,
                        qsmc_stream_t stream);

/* The kernel-density cross term of est_kl_divergence / SMCUpdater's resampling divergences (distributions.py:466-487,
 * smc.py:506-542; distances metrics.py:72-106):
 *     *out_host = sum_i (w_i / norm_p) log( sum_j (v_j / norm_q) phi(|| scale o (x_i - y_j) ||_2) ),
 * phi the standard normal pdf, scale[q] = sqrt(Q_q) / delta on the host (d doubles); w or v NULL: all-ones weights.
 * The divergence is -est_entropy(p) - *out_host / delta.  O(n m d) work, nothing of size n x m is stored. */
int qsmc_kde_cross_entropy(qsmc_handle_t h, const double *x, int64_t ldx, int64_t n, const double *w, double norm_p,
                           const double *y, int64_t ldy, int64_t m, const double *v, double norm_q, int32_t d,
                           const double *scale_host, double *out_host, qsmc_stream_t stream);

/* Materialise normalised weights: w_out[i] = w_in[i] / norm (particle_weights property). */
int qsmc_normalize_weights(qsmc_handle_t h, const double *w_in, double *w_out, int64_t n,
                           double norm, qsmc_stream_t stream);

/* w[i] = value (reset smc.py:307; resamplers.py:390). */
int qsmc_fill(qsmc_handle_t h, double *w, int64_t n, double value, qsmc_stream_t stream);

/* ---- weighted moments (distributions.py:337-399, utils.py:216-287; a11-a12) --------------- */
/* out_host / out_dev: [ sum w~, sum w~ x_m (d), sum w~ x_m x_n for m <= n row-major (d(d+1)/2) ]
 * with w~ = w / norm.  Synchronises if out_host != NULL.  d <= 4: VALU; 4 < d <= 16 and 16 < d <= QSMC_MAX_D_WIDE: the
 * contraction X diag(w) X^T on the f64 matrix cores (k_moments_mfma; k_moments_wide in 16 x 16 blocks). */
int qsmc_moments(qsmc_handle_t h, const double *x, int64_t ldx, int64_t n, int32_t d,
                 const double *w, double norm, double *out_dev, double *out_host,
                 qsmc_stream_t stream);

/* utils.py:593-607 sqrtm_psd on the HOST (d x d row-major): S = scale * V sqrt(max(lambda,0)) V^T,
 * err = ||S S / scale^2 - A||_F.  Cyclic Jacobi; no device work. */
int qsmc_sqrtm_psd(const double *A, int32_t d, double scale, double *S_out, double *err_out);

/* ---- Liu-West resampling (resamplers.py:256-392; a15) ------------------------------------ */
/* cdf[i] = sum_{j<=i} w[j]/norm   (np.cumsum, resamplers.py:308) */
int qsmc_cumsum(qsmc_handle_t h, const double *w, int64_t n, double norm, double *cdf,
                qsmc_stream_t stream);

/* js[i] = min(#{j : cdf[j] <= u[i]}, n_in - 1)  (searchsorted side='right', :318-321, clamped as
 * distributions.py:330-333 does; the reference raises IndexError on the unclamped overflow, Q2). */
int qsmc_lw_ancestors(qsmc_handle_t h, const double *cdf, int64_t n_in,
                      const double *u, int64_t n_out, int64_t *js, qsmc_stream_t stream);

/* mus[m][i] = a * x_in[m][js[i]] + (1 - a) * mean[m]   (:325).  mean is HOST (d).  d <= QSMC_MAX_D_WIDE. */
int qsmc_lw_centres(qsmc_handle_t h, const double *x_in, int64_t ldx_in, int32_t d,
                    const int64_t *js, int64_t n_out, double a, const double *mean,
                    double *mus, int64_t ld_mus, qsmc_stream_t stream);

/* One postselection round (:327-372):  for r in [0, k):
 *     dst = idxs ? idxs[r] : r
 *     c   = centre_by_idx ? dst : r          (r = the reference's `mus[:k]` truncation, quirk Q1)
 *     x_out[:, dst] = mus[:, c] + S @ z[:, r]        z is DEVICE [d][k] (param-major, :332)
 *     valid_out[r]  = model.are_models_valid(x_out[:, dst])   (1 if !postselect)
 * S is HOST d x d row-major (already scaled by h, :300).  d > QSMC_MAX_D (tomography): no validity test, valid_out = 1. */
int qsmc_lw_perturb(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect,
                    const double *mus, int64_t ld_mus, const int64_t *idxs, int64_t k,
                    int32_t centre_by_idx, const double *S, const double *z, int64_t ldz,
                    double *x_out, int64_t ldx_out, uint8_t *valid_out, qsmc_stream_t stream);

/* Device-RNG resample straight from the (unnormalised) weights (Philox4x32-10, deterministic per
 * (seed, epoch, slot)): multinomial ancestors -> Liu-West centre -> Box-Muller kick -> validity; an
 * invalid particle redraws ancestor and kick up to maxiter times.  w == NULL: all-ones weights.
 * *n_failed_host = particles still invalid (synchronises); NULL stays asynchronous
 * (qsmc_last_resample_failed).
 * For 16384 <= n_out < 2^32 and n_in <= 3.3e7 the BUCKETED sampler runs: exact
 * multinomial counts per 4096-particle chunk (independent Poisson draws per chunk brought to the exact
 * total by a short categorical top-up: the law of the reference's n_out searches of the CDF,
 * resamplers.py:308-316, without n_out uniforms), then one workgroup per chunk scans ITS weights in LDS,
 * so the CDF is never written to HBM; outputs come ordered by ancestor chunk (particles are
 * exchangeable; same joint law).  The global CDF is materialised only if a particle needs a global
 * redraw.  Otherwise: CDF + one binary search per particle.
 * 16 < d <= QSMC_MAX_D_WIDE (QSMC_MODEL_TOMOGRAPHY; no validity test): the same ancestors -- k_bucket_anc16 where the
 * bucketed sampler applies, else the direct search -- and the kicks S z on the f64 matrix cores (k_kick_wide); mean and S
 * are copied to the device by the call.  Canonicalize is a call of its own (qsmc_tomo_canonicalize2). */
int qsmc_lw_resample_philox(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect,
                            const double *x_in, int64_t ldx_in, int64_t n_in, int32_t d,
                            const double *w, double norm, double a, const double *mean, const double *S,
                            int64_t n_out, uint64_t seed, uint64_t epoch, int32_t maxiter,
                            double *x_out, int64_t ldx_out, int64_t *n_failed_host,
                            qsmc_stream_t stream);

/* (resamplers.py:308-316, the cumsum the reference takes before searching)
 * Every qsmc_update_fused also leaves the sum of the new weights per kernel tile (2048 particles) in the
 * handle and bumps a generation counter (qsmc_update_token).  A caller that KNOWS the weights it is about
 * to resample are exactly the w_out of the update with that token -- untouched since -- may say so with
 * qsmc_lw_use_update_sums(token) right before qsmc_lw_resample_prepare / qsmc_lw_resample_philox: the
 * resampler then forms its chunk sums from those tile sums instead of reading the weights once more
 * (one 8 B/particle pass saved).  A stale or wrong token is ignored; any call that rewrites weights
 * invalidates the sums.  Chunk edges then differ from the plain path by rounding only (same law). */
int qsmc_update_token(qsmc_handle_t h, uint64_t *token_out);
int qsmc_lw_use_update_sums(qsmc_handle_t h, uint64_t update_token);

/* Queue the weight-only prefix of the NEXT qsmc_lw_resample_philox call (chunk sums, offsets, and for
 * the bucketed sampler the multinomial chunk counts and work-item plan): none of it needs the mean or
 * the covariance square root, so a caller can launch it the moment its n_ess test fails
 * (smc.py:273-277) and form mean / cov / sqrtm (resamplers.py:266-300) on the host while the GPU is
 * already busy.  The following qsmc_lw_resample_philox with the SAME (w, n_in, norm, n_out, seed,
 * epoch, stream) then starts at the sampling kernel; any other arguments, or any call in between that
 * changes weights, simply redo the prefix -- results are identical either way. */
int qsmc_lw_resample_prepare(qsmc_handle_t h, const double *w, int64_t n_in, double norm, int64_t n_out,
                             uint64_t seed, uint64_t epoch, qsmc_stream_t stream);

/* (smc.py:263-277: the n_ess test that decides on a resample, taken on the device)
 * enabled != 0: every later qsmc_update_fused with host-visible statistics queues that same weight-only prefix for
 * (n_out, seed, epoch) right behind its reducing kernel, gated ON THE DEVICE by the test the host is about to make
 * -- (sum w')^2 / sum w'^2 < ess_below, no negative weight, |sum w'| not below machine epsilon -- with the same IEEE
 * operations on the same sums.  A closed gate costs a ~3 us launch that leaves at once, inside the host's round
 * trip; an open one has the counts and the plan under way while the host still waits for the update's statistics.
 * qsmc_lw_resample_prepare / qsmc_lw_resample_philox with matching arguments (and the caller's qsmc_lw_use_update_sums
 * vouching for the weights) then find the prefix done; with anything different they redo it as before -- the device's
 * decision never replaces the caller's, it only starts early the work the caller is about to ask for.  Same counts,
 * same particles either way.  Re-arm after every resample (the epoch moves on); enabled = 0 stops it. */
int qsmc_lw_arm_prefix(qsmc_handle_t h, int32_t enabled, double ess_below, int64_t n_out, uint64_t seed,
                       uint64_t epoch);
/* how many prefixes were queued that way on this handle, and how many resamples found theirs done */
int qsmc_lw_prefix_stats(qsmc_handle_t h, int64_t *n_queued, int64_t *n_adopted);

/* TomographyModel.canonicalize folded into the resample (smc.py:529 runs it on the fresh cloud; tomography/models.py:149-209):
 * the NEXT qsmc_lw_resample_philox on this handle -- a d = 16 TOMOGRAPHY model on the bucketed path -- classifies every
 * new particle while it still holds it in registers (LDL^H pivot test: a positive-definite rho only needs x / (x_0 sqrt dim),
 * done in the same store), lists the others and runs the eigenvalue-clamping pass (k_tomo_canon_list) on the list:
 * the result is what qsmc_tomo_canonicalize2(basis, 4, basis_kind, x_out, ...) would leave, bit for bit, without the
 * pass that re-read and re-wrote the whole cloud.  One-shot: consumed (or dropped, if it does not apply: status
 * QSMC_ERR_UNSUPPORTED from the resample) by that call; any call that changes weights clears it. */
int qsmc_lw_fuse_canonicalize(qsmc_handle_t h, const double *basis, int32_t dim, int32_t basis_kind,
                              int32_t allow_subnormalized);
/* Does a d = 16 resample of this shape take the split sampler that can fold canonicalize in (the rule the library itself
 * applies: the bucketed path -- at most 8192 chunks of 4096 source particles, at least 4 chunks' worth of outputs)?
 * 1 / 0.  A caller asks this instead of restating the rule. */
int qsmc_lw_can_fuse_canonicalize(int32_t d, int64_t n_in, int64_t n_out);

/* Grow every scratch buffer that updates of a cloud of n_in particles (d parameters) and its Liu-West resample into n_out
 * particles would otherwise grow at first use (chunk tables, plan, CDF, partial sums ...): called once when a cloud is
 * set up, so that the first resample of a process costs what the hundredth does.  (Round 3: a driver run whose 5
 * warm-up data triggered no resample timed 3.5 ms for 20 steps that take 1.6 ms.)  Idempotent; allocates, never frees
 * below what is in use. */
int qsmc_reserve(qsmc_handle_t h, int64_t n_in, int64_t n_out, int32_t d);

/* d = 16 resamples queued by qsmc_step whose covariance square root was formed on the device (kernels/sqrtm.hpp), and how
 * many of those the host's own square root confirmed bit for bit (only those are adopted). */
int qsmc_step_sqrt_stats(qsmc_handle_t h, int64_t *n_device, int64_t *n_agreed);


/* Postselection without global redraws (resamplers.py:341-372) for models whose constraint bites at every resample (RB):
 * n_expected = how many first tries of THIS cloud's previous resample failed postselection (qsmc_last_resample_redraws
 * after the update that followed it).  One-shot, consumed by the next qsmc_lw_resample_philox: with n_expected > 0 and
 * d = 3 or 4 the ordered sampler also produces ~1.25 n_expected spare proposals (ancestor ~ w, kick: exactly what a
 * redraw draws) while each chunk's CDF is in LDS anyway, and a failed slot is served spares -- in an order that does not
 * depend on their values -- until one is valid; only slots the bank could not serve go to the global-CDF redraw kernel.
 * Same law; different particles than with n_expected = 0 (the redraws consume other Philox blocks); deterministic. */
int qsmc_lw_expect_redraws(qsmc_handle_t h, int64_t n_expected);

/* qsmc_lw_resample_philox with n_failed_host == NULL does not synchronise: the failed-particle count
 * is written to pinned host memory by the stream; read it here once `stream` has been synchronised by
 * any later call (synchronize = 0), or force the wait (synchronize = 1). */
int qsmc_last_resample_failed(qsmc_handle_t h, int64_t *n_failed_out, int32_t synchronize,
                              qsmc_stream_t stream);
/* How many outputs of the latest resample asked for a global redraw (their first ancestor's kick failed postselection,
 * resamplers.py:341-372), as published with the latest host-visible reduction after it: a diagnostic. */
int qsmc_last_resample_redraws(qsmc_handle_t h, int64_t *n_redraws_out);

/* Sharded resampling: THIS rank produces the finished Liu-West particles for every
 * destination rank -- the kick needs only the ancestor and the (already all-reduced) global mean /
 * covariance, so nothing about the destination enters -- and they leave by one all-to-all.
 * dest_counts[r] (HOST, n_dest <= 16) = how many particles rank r takes from this shard (column of
 * the shared count matrix); rows_out is AoS [sum(dest_counts)][d], grouped by destination rank.
 * Inside each group the rows are an even round-robin deal of the chunk-sorted sample, so shards stay
 * exchangeable.  `norm` is this shard's own weight sum (its local CDF ends at 1).  A postselection
 * retry redraws its ancestor from this shard (exact for exchangeable shards). */
int qsmc_lw_resample_philox_sharded(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect,
                                    const double *x_in, int64_t ldx_in, int64_t n_in, int32_t d,
                                    const double *w, double norm, double a, const double *mean, const double *S,
                                    const int64_t *dest_counts, int32_t n_dest, uint64_t seed,
                                    uint64_t epoch, int32_t maxiter, double *rows_out,
                                    int64_t *n_failed_host, qsmc_stream_t stream);

/* Device-RNG uniform-box prior (distributions.py:792-827 + :1304-1350 postselection):
 * x[m][i] = lo[m] + U * (hi[m] - lo[m]), redrawn in-thread while invalid (<= maxiter). */
int qsmc_prior_uniform_philox(qsmc_handle_t h, const qsmc_model_t *model, int32_t postselect,
                              const double *lo, const double *hi, int32_t d, int64_t n,
                              uint64_t seed, uint64_t epoch, int32_t maxiter,
                              double *x_out, int64_t ldx_out, int64_t *n_failed_host,
                              qsmc_stream_t stream);

/* ---- tomography canonicalize (tomography/models.py:149-209; a10) -------------------------- */
/* Random-walk time step between data (Model.update_timestep, smc.py:447-449; RandomWalkModel /
 * GaussianRandomWalkModel, derived_models.py:693-741, 743-963): x[m][i] += scale[m] * z in place; rows
 * with scale[m] == 0 do not move.  z != NULL: device array, row r of the WALKING parameters (in index
 * order) at z + r * ldz -- steps the host drew (the reference's np.random.normal stream in parity mode,
 * or any step distribution).  z == NULL: standard normals from Philox4x32-10 keyed by (seed, epoch).
 * `scale` is a HOST array of d doubles (std per parameter times the experiment's scale multiplier).
 * d <= QSMC_MAX_D_WIDE. */
int qsmc_random_walk(qsmc_handle_t h, double *x, int64_t ldx, int64_t n, int32_t d, const double *scale,
                     const double *z, int64_t ldz, uint64_t seed, uint64_t epoch, qsmc_stream_t stream);

/* basis: DEVICE complex128 (d, dim, dim) row-major as interleaved (re, im), d = dim*dim, dim = 2 .. 8
 * (dim 5 .. 8: two passes -- an LDL^H pivot test sorts out the positive-definite particles, the rest take the Jacobi form).
 * In place: clamp negative eigenvalues of rho(x), then x /= x_0 sqrt(dim) unless allow_subnormalized. */
int qsmc_tomo_canonicalize(qsmc_handle_t h, const double *basis, int32_t dim,
                           double *x, int64_t ldx, int64_t n, int32_t allow_subnormalized,
                           qsmc_stream_t stream);
/* The same with the basis NAMED: basis_kind = QSMC_BASIS_PAULI says `basis` is the reference's n-qubit Pauli basis
 * (pauli_basis(nq), tomography/bases.py:137-154: tensor products of (I, X, Y, Z) / sqrt 2, first qubit slowest);
 * for dim = 4 the library then contracts with the basis' four non-zero entries per element instead of the dense
 * (16, 4, 4) tensor (`basis` may be NULL).  QSMC_BASIS_DENSE: any orthonormal Hermitian basis, as above. */
#define QSMC_BASIS_DENSE 0
#define QSMC_BASIS_PAULI 1
int qsmc_tomo_canonicalize2(qsmc_handle_t h, const double *basis, int32_t dim, int32_t basis_kind,
                            double *x, int64_t ldx, int64_t n, int32_t allow_subnormalized,
                            qsmc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* QSMC_H */
