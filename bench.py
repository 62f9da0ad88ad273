#!/usr/bin/env python3
"""bench.py -- particle-updates/sec of the SMC hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N --steps K --warmup W          (no launcher: spawns its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE config 2 = SURVEY C2): SimplePrecessionModel, 1e7 particles PER GPU, fp64,
LiuWestResampler(a=0.98), resample_thresh 0.5, prior U[0,1], true omega = 0.3, experiment schedule
t_k = (9/8)^k (k = 0..199, wrapping with a prior reset), outcomes simulated once on the host from a fixed
seed (they do not depend on N).  A "step" is one `SMCUpdater.update(outcome_k, t_k)` -- exactly what the
reference's perf_test times (perf_testing.py:250-251) -- INCLUDING any resample it triggers.  The cloud is
resident in HBM before the timed region; device RNG (Philox).

One JSON line on rank 0 with `value` = N_total * K / wall, plus
  roofline            the fused update kernel's achieved algorithmic HBM bytes/s (24 B/particle) from HIP-event
                      kernel durations measured inside the timed region (+ a census of every kernel kind taken right
                      after it, all launches timed, so that the numbers do not hang on two launches at --steps 20);
  roofline_beyond_l3  the same kernel at N = 1e8 (2.4 GB working set: past the 256 MB Infinity Cache);
  other_configs       BASELINE configs 3, 4 (per-GPU share) and 5 (per-GPU share) with their SURVEY 8(d) schedules:
                      p-u/s, and each one's kernels against their 16 + 8 d bytes / particle;
  other_paths         SURVEY 8(f): `batch_update` at resample_interval 5 and 8 and one `bayes_risk` call over 26 outcomes,
                      each with its kernel (k_update_multi, k_hyp_sums) against its algorithmic bytes;
  cpu_baseline        the C / OpenMP restatement of the reference (oracle/cpu_port.c, pinned to the reference's golden
                      trajectories by tests/test_cpu_port.py) on this box's host cores, 1 thread and all cores, on the
                      same cloud size and the first data of the same schedule;
  transports          (sharded runs: --gpus N > 1, or --force-comm) the same K steps under each transport of the
                      per-datum reduction: "shm" (host shared memory, the default on one node: `value` is this pass)
                      and "rccl" (the library's own RCCL all-gather on the launch stream) with `ranks_in_comm` read
                      back from the communicator (ncclCommCount).
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "python-qinfer_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
N_SCHEDULE = 200
TAGS = {0: "update", 1: "sample", 2: "update_ones", 3: "canon_classify", 4: "canon_list", 5: "moments", 6: "counts",
        8: "ancestors", 10: "update_multi", 11: "hyp_sums", 12: "canon_build", 13: "canon_expand"}


def schedule():
    ts = (9.0 / 8.0) ** np.arange(N_SCHEDULE, dtype=np.float64)
    rs = np.random.RandomState(0)
    pr0 = np.cos(0.3 * ts / 2) ** 2
    outcomes = (rs.random_sample(N_SCHEDULE) >= pr0).astype(np.int64)
    return ts, outcomes


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(n_particles, n_data, gpu_same_sample):
    """The C / OpenMP port of the reference path (oracle/cpu_port.c) on this host: same cloud size, the first
    `n_data` data of the same schedule, Philox draws keyed by particle (so the thread count does not change the
    algorithm), timed at 1 thread and at all cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_port as cp
    ts, outcomes = schedule()
    x0 = np.random.RandomState(1).random_sample((n_particles, 1))
    cores = usable_cores()
    # which thread count is "all cores" for this box?  A container may see 256 logical CPUs and be allowed far
    # fewer by its cgroup, or the memory system may stop scaling long before the core count: a short probe (3 data,
    # no resample) at a few thread counts picks the fastest, and the full sample runs with that
    cand = sorted({c for c in (8, 16, 32, 64, 128, cores) if c <= cores} | {cores})
    probe = {}
    for th in cand:
        r = cp.smc_run(cp.PRECESSION, x0, outcomes[:3], t=ts[:3], rng_mode=1, seed=0, threads=th)
        probe[th] = n_particles * 3 / r["wall_s"]
    best = max(probe, key=probe.get)
    runs = {}
    for label, th in (("all_cores", best), ("one_thread", 1)):
        r = cp.smc_run(cp.PRECESSION, x0, outcomes[:n_data], t=ts[:n_data], rng_mode=1, seed=0, threads=th)
        if r["rc"] != 0:
            return {"error": "cpu port returned %d" % r["rc"]}
        runs[label] = {"value": n_particles * n_data / r["wall_s"], "threads": int(r["threads"]), "wall_s": r["wall_s"],
                       "update_s": r["update_s"], "resample_s": r["resample_s"], "resamples": int(r["resample_count"]),
                       "posterior_mean": float(r["mean"][0])}
    runs["all_cores"]["thread_probe_updates_only"] = {str(k): v for k, v in probe.items()}
    allc = runs["all_cores"]
    return {"value": allc["value"], "unit": "particle-updates/s", "cores": allc["threads"], "kind": "port",
            "host_cpu_count": os.cpu_count(), "usable_cores": cores,
            "sample": "oracle/cpu_port.c (C/OpenMP restatement of smc.py:388-457 + resamplers.py:256-392, pinned to the "
                      "reference's golden trajectories in tests/test_cpu_port.py), SimplePrecession, N=%d, first %d data "
                      "of the headline schedule (%d resamples), Philox draws keyed by particle; %.1f s wall on %d threads, "
                      "%.1f s on 1 thread" % (n_particles, n_data, allc["resamples"], allc["wall_s"], allc["threads"],
                                              runs["one_thread"]["wall_s"]),
            "all_cores": allc, "one_thread": runs["one_thread"], "gpu_same_sample": gpu_same_sample}


def usable_cores():
    """CPUs this process may actually use: the affinity mask, capped by a cgroup CPU quota if one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        n = os.cpu_count() or 1
    try:                                             # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:  # noqa: BLE001
        try:                                         # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:  # noqa: BLE001
            pass
    return n


def load_traffic():
    """HBM bytes per launch from committed rocprofv3 --pmc passes (profiles/hbm_traffic.json), if any."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        return json.load(open(path))
    except Exception:  # noqa: BLE001
        return {}


def kernel_table(ms, tags):
    out = {}
    for t, name in TAGS.items():
        sel = ms[tags == t]
        if len(sel):
            out[name] = {"launches": int(len(sel)), "avg_us": float(sel.mean()) * 1e3, "min_us": float(sel.min()) * 1e3}
    return out


def frac_entry(kernel, avg_us, n_bytes, launches, extra=None):
    ach = n_bytes / (avg_us * 1e-6) / 1e9
    e = {"kernel": kernel, "avg_kernel_us": avg_us, "timed_launches": launches, "algorithmic_bytes_per_launch": n_bytes,
         "achieved": ach, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS}
    if extra:
        e.update(extra)
    return e


# ------------------------------------------------------------------------------------------------ other configs
def cached_prior(qi, dist):
    """A host-sampled prior drawn once per cloud size (the Ginibre prior of 1.25e6 states takes seconds on the
    host; the bench resets the updater several times and prior sampling is outside every timed region)."""
    class Cached(qi.Distribution):
        n_rvs = dist.n_rvs
        _memo = {}

        def sample(self, n=1):
            if n not in self._memo:
                self._memo[n] = dist.sample(n)
            return self._memo[n]
    return Cached()


def other_config_specs(qi):
    """BASELINE configs 3, 4 (share) and 5 (share) as concrete synthetic inputs."""
    specs = []
    rs = np.random.RandomState(0)
    K = 60
    # C3: BinomialModel(SimplePrecessionModel) n_meas = 25, N = 1e7 (derived_models.py:314-329)
    m = qi.BinomialModel(qi.SimplePrecessionModel())
    eps, outs = [], []
    for k in range(K):
        ep = np.empty((1,), dtype=m.expparams_dtype)
        ep['x'] = (9 / 8) ** k
        ep['n_meas'] = 25
        eps.append(ep)
        outs.append(int(rs.binomial(25, np.sin(0.3 * (9 / 8) ** k / 2) ** 2)))
    specs.append(dict(key="config3_binomial_precession", model=m, n=10_000_000, d=1,
                      prior=lambda: qi.UniformDistribution([0, 1]), eps=eps, outs=outs,
                      workload="BinomialModel(SimplePrecessionModel) n_meas=25, 1e7 particles, t_k=(9/8)^k",
                      update_kernel="k_update_fused<BINOMIAL_PRECESSION,2,false>", sampler="k_bucket_sample<1,512>"))
    # C4 per-GPU share: RandomizedBenchmarkingModel d = 3, N = 1.25e7 (rb.py:178-195), prior as simple_est_rb
    m = qi.RandomizedBenchmarkingModel()
    eps, outs = [], []
    for k in range(K):
        ep = np.empty((1,), dtype=m.expparams_dtype)
        ep['m'] = 1 + 5 * k
        eps.append(ep)
        outs.append(int(rs.random_sample() >= 1 - (0.3 * 0.95 ** (1 + 5 * k) + 0.5)))
    specs.append(dict(key="config4_share_rb", model=m, n=12_500_000, d=3,
                      prior=lambda m=m: qi.PostselectedDistribution(qi.UniformDistribution([[0.8, 1], [0, 1], [0, 1]]), m),
                      eps=eps, outs=outs,
                      workload="RandomizedBenchmarkingModel (p, A, B), 1.25e7 particles (1e8 / 8 GPUs), m_k = 1 + 5k",
                      update_kernel="k_update_fused<RB,1,false>", sampler="k_bucket_sample_ordered<3,512>"))
    # C5 per-GPU share: 2-qubit TomographyModel d = 16, N = 1.25e6, Ginibre prior, random Pauli measurements
    basis = qi.tomography.pauli_basis(2)
    m = qi.TomographyModel(basis)
    np.random.seed(0)
    gin = qi.GinibreDistribution(basis)
    true = gin.sample(1)[0]
    eps, outs = [], []
    for k in range(K):
        ep = np.zeros((1,), dtype=m.expparams_dtype)
        p = rs.randint(1, 16)
        ep['meas'][0, 0] = 1
        ep['meas'][0, p] = 1
        eps.append(ep)
        outs.append(int(rs.random_sample() < np.clip(true[0] + true[p], 0, 1)))
    gin_cached = cached_prior(qi, gin)
    specs.append(dict(key="config5_share_tomography", model=m, n=1_250_000, d=16, prior=lambda: gin_cached, eps=eps, outs=outs,
                      workload="2-qubit TomographyModel (15 free params), 1.25e6 particles (1e7 / 8 GPUs), Ginibre prior, "
                               "random Pauli measurements",
                      update_kernel="k_update_tomo<NNZ=2,false>", sampler="k_bucket_anc16<512> + k_bucket_kick16",
                      # a Pauli measurement (I + P) / 2 = e_0 + e_P has two nonzero entries: the likelihood reads w and those
                      # two of the 16 rows, writes w' (the dense form, 16 + 8 d = 144 B, multiplies 14 rows by zero)
                      update_bpp=32, update_note="sparse measurement vector: 16 + 8 nnz = 32 B per particle "
                                                 "(dense form: 144 B; QSMC_TOMO_DENSE_UPDATE=1)"))
    # C4 and C5 at their FULL sizes on ONE GPU (BASELINE quotes them on 8; both fit one MI355X's HBM: 3.2 GB and 1.36 GB of
    # cloud): the same schedules as the shares above, so the two entries show what the per-datum fixed cost does to a share.
    # C4's resample runs in segments (more than 8192 x 4096 particles); C5's prior is drawn on the device
    # (GinibreDistribution.sample_device: 1e7 states on the host take half a minute).
    c4 = specs[-2]
    specs.append(dict(c4, key="config4_full_rb_1gpu", n=100_000_000,
                      workload="RandomizedBenchmarkingModel (p, A, B), 1e8 particles on ONE GPU (BASELINE config 4's total), "
                               "m_k = 1 + 5k"))
    c5 = specs[-2]
    specs.append(dict(c5, key="config5_full_tomography_1gpu", n=10_000_000, prior=lambda: qi.GinibreDistribution(basis, device=True),
                      workload="2-qubit TomographyModel (15 free params), 1e7 particles on ONE GPU (BASELINE config 5's "
                               "total), Ginibre prior (device draw), random Pauli measurements"))
    # Not a BASELINE config -- the widening of round 6: THREE-qubit tomography (d = 64; the reference's TomographyModel takes
    # any dim, tomography/models.py:82-226) on the wide kernels (csrc/kernels/wide.hpp), N = 5e5 (256 MB of cloud), Ginibre
    # prior, 150 random Pauli measurements (a schedule long enough for resamples at this dimension)
    basis3 = qi.tomography.pauli_basis(3)
    m = qi.TomographyModel(basis3)
    gin3 = qi.GinibreDistribution(basis3)
    true3 = gin3.sample(1)[0]
    eps, outs = [], []
    for k in range(150):
        ep = np.zeros((1,), dtype=m.expparams_dtype)
        p = rs.randint(1, 64)
        ep['meas'][0, 0] = np.sqrt(8) / 2                    # (I + P) / 2 = (sqrt 8 / 2)(B_0 + B_P), B_a = P_a / sqrt 8
        ep['meas'][0, p] = np.sqrt(8) / 2
        eps.append(ep)
        outs.append(int(rs.random_sample() < np.clip(ep['meas'][0] @ true3, 0, 1)))
    gin3_cached = cached_prior(qi, gin3)
    specs.append(dict(key="widening_tomography_3q", model=m, n=500_000, d=64, prior=lambda: gin3_cached, eps=eps, outs=outs,
                      workload="3-qubit TomographyModel (63 free params, d = 64), 5e5 particles, Ginibre prior, 150 random Pauli "
                               "measurements; NOT a BASELINE config: the reference's any-dim tomography on the wide kernels",
                      update_kernel="k_update_tomo_wide<2,false>", sampler="k_bucket_anc16<512> + k_kick_wide<4,false>",
                      sampler_split_label="k_bucket_anc16<512> + k_kick_wide<4> (S z on v_mfma_f64_16x16x4, S from device memory)",
                      canon_label="k_gemm_wide<4,0> + k_tomo_ldl_wide<8> + k_tomo_jacobi_wide<8> + k_gemm_wide<4,1> (rho packed on the matrix cores; one-sided Jacobi, no eigenvectors)", moments_label="k_moments_wide<4>",
                      update_bpp=32, update_note="sparse measurement vector: 16 + 8 nnz = 32 B per particle (dense: 528 B)"))
    # (not a BASELINE config, not in the default run: `--only extra_binomial_rb` -- the model simple_est_rb builds)
    m = qi.BinomialModel(qi.RandomizedBenchmarkingModel())
    eps, outs = [], []
    for k in range(K):
        ep = np.empty((1,), dtype=m.expparams_dtype)
        ep['m'] = 1 + 5 * k
        ep['n_meas'] = 25
        eps.append(ep)
        outs.append(int(rs.binomial(25, 0.3 * 0.95 ** (1 + 5 * k) + 0.5)))
    specs.append(dict(key="extra_binomial_rb", model=m, n=12_500_000, d=3, extra=True,
                      prior=lambda m=m: qi.PostselectedDistribution(qi.UniformDistribution([[0.8, 1], [0, 1], [0, 1]]), m),
                      eps=eps, outs=outs, workload="BinomialModel(RandomizedBenchmarkingModel) n_meas=25, 1.25e7 particles",
                      update_kernel="k_update_fused<BINOMIAL_RB,1,false>", sampler="k_bucket_sample_ordered<3,512>"))
    return specs


def run_other_config(qi, eng, torch, spec, warmup, comm=None, n_override=None, sync=None):
    """One BASELINE config over its 60-datum schedule.  `comm`: the cloud is one shard of `comm.world_size` (every rank
    calls this with the same spec; the timed loops sit between `sync()` barriers and the wall time is the slowest
    rank's); `n_override`: particles per rank instead of spec["n"] (the shared-GPU control-flow check)."""
    n, d = (n_override or spec["n"]), spec["d"]
    eps, outs = spec["eps"], spec["outs"]
    world = 1 if comm is None else comm.world_size
    if sync is None:
        sync = torch.cuda.synchronize
    if comm is not None:
        np.random.seed(1000 + comm.rank)                        # (host-sampled priors: a different draw per shard)
    import gc
    # (the collector stays off through the timed loops, as in `timeit` and in the headline's timed pass; collected here,
    #  before anything is on the GPU)
    gc.collect()
    gc.disable()
    try:                                                        # (a failure in here must not leave the collector off)
        upd = qi.SMCUpdater(spec["model"], n, spec["prior"](), device_rng=True, seed=0, comm=comm)
        for k in range(min(warmup, len(eps))):                  # untimed: allocator growth, first-launch costs
            upd.update(outs[k], eps[k])
        upd.resample()
        upd.update(outs[0], eps[0])
        upd.reset()
        upd._resample_count = 0
        sync()
        eng.set_profiling(0 if os.environ.get("QSMC_BENCH_NO_EVENTS") else 1)  # (no events: for kernel-trace gap profiles)
        t0 = time.perf_counter()
        for k in range(len(eps)):
            upd.update(outs[k], eps[k])
        sync()
        wall = time.perf_counter() - t0
        ms, tags = eng.profile_read()
        eng.set_profiling(0)
        # the same loop again without kernel events: the throughput figure (events drain the queue around each launch)
        upd.reset()
        rc0 = upd.resample_count
        rb0 = 0 if comm is None else comm.n_rebalances
        sync()
        t0 = time.perf_counter()
        for k in range(len(eps)):
            upd.update(outs[k], eps[k])
        sync()
        wall = time.perf_counter() - t0
    finally:
        gc.enable()
    if world > 1:
        wt = torch.tensor([wall], dtype=torch.float64, device="cuda" if comm.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(wt, op=torch.distributed.ReduceOp.MAX)
        wall = float(wt.item())
    kt = kernel_table(ms, tags)
    K = len(eps)
    out = {"workload": spec["workload"], "particles": n, "d": d, "steps": K, "resamples": upd.resample_count - rc0,
           "value": n * world * K / wall, "unit": "particle-updates/s", "ms_per_step": wall / K * 1e3,
           "posterior_mean_head": [float(v) for v in upd.est_mean()[:3]]}
    if comm is not None:
        out.update({"particles_per_rank": n, "particles": n * world, "ranks": world,
                    "per_datum_collective": comm.transport_name, "rebalances": comm.n_rebalances - rb0,
                    "resample_path": getattr(comm, "last_resample_path", None)})
    if "update" in kt:
        bpp = spec.get("update_bpp", 16 + 8 * d)
        extra = {"bytes_per_particle": bpp}
        if "update_note" in spec:
            extra["note"] = spec["update_note"]
        out["update_kernel"] = frac_entry(spec["update_kernel"], kt["update"]["avg_us"], bpp * n,
                                          kt["update"]["launches"], extra)
    if "sample" in kt:
        if "ancestors" in kt:
            # d = 16: the sampler is two kernels (ancestors, then the kicks with canonicalize's classify pass folded in)
            us = kt["sample"]["avg_us"] + kt["ancestors"]["avg_us"]
            out["resample_kernel"] = frac_entry(spec.get("sampler_split_label", "k_bucket_anc16<512> + k_bucket_kick16 (classify fused)"),
                                                us, (8 + 16 * d) * n,
                                                kt["sample"]["launches"],
                                                {"bytes_per_particle": 8 + 16 * d, "ancestors_us": kt["ancestors"]["avg_us"],
                                                 "kick_us": kt["sample"]["avg_us"]})
        else:
            # beyond the resampler's segment limit a resample is one sampler launch per SEGMENT: a launch's bytes are its
            # segment's
            segs = -(-n // qi.LiuWestResampler._segment_limit)
            extra = {"bytes_per_particle": 8 + 16 * d}
            if segs > 1:
                extra["segments_per_resample"] = segs
            out["resample_kernel"] = frac_entry(spec["sampler"], kt["sample"]["avg_us"], (8 + 16 * d) * n // segs,
                                                kt["sample"]["launches"], extra)
    if "canon_classify" in kt:
        zero = {"avg_us": 0.0}
        cus = (kt["canon_classify"]["avg_us"] + kt.get("canon_list", zero)["avg_us"] + kt.get("canon_build", zero)["avg_us"]
               + kt.get("canon_expand", zero)["avg_us"])
        extra = {"bytes_per_particle": 16 * d, "classify_us": kt["canon_classify"]["avg_us"],
                 "canon_list_us": kt.get("canon_list", zero)["avg_us"]}
        if "canon_build" in kt:                 # the wide form: two products on the matrix cores around the pivot test / Jacobi
            extra.update({"build_us": kt["canon_build"]["avg_us"], "expand_us": kt.get("canon_expand", zero)["avg_us"]})
        out["canonicalize"] = frac_entry(spec.get("canon_label", "k_tomo_classify<4> + k_tomo_canon_list<4>"), cus, 16 * d * n,
                                         kt["canon_classify"]["launches"], extra)
    elif "canon_list" in kt:
        # classify rides in the kick kernel (no pass of its own): what is left of canonicalize is the list pass over the
        # particles whose rho is not positive definite (about a third: read + write 16 rows of those)
        out["canonicalize"] = {"kernel": "k_tomo_canon_list_fast (eigenvector-free clamp; classify fused into k_bucket_kick16)",
                               "avg_kernel_us": kt["canon_list"]["avg_us"], "timed_launches": kt["canon_list"]["launches"],
                               "classify_us": 0.0, "canon_list_us": kt["canon_list"]["avg_us"]}
    if "moments" in kt:
        out["moments_kernel"] = frac_entry(spec.get("moments_label", "k_moments_mfma"), kt["moments"]["avg_us"], (8 + 8 * d) * n,
                                           kt["moments"]["launches"], {"bytes_per_particle": 8 + 8 * d})
    del upd
    torch.cuda.empty_cache()
    return out


def other_paths(qi, eng, torch, n=10_000_000):
    """SURVEY 8(f) rows 1-2 on the headline cloud size: `batch_update` (smc.py:459-487; data between two n_ess tests
    applied in ONE pass over the cloud, k_update_multi) at resample_interval 5 and 8, and one `bayes_risk` call
    (smc.py:553-611) over the 26 outcomes of a Binomial(n_meas = 25) experiment (k_hyp_sums: every outcome's
    hypothetical normalisation and moments in one pass, nothing n_outcomes x N stored).  Kernel times from HIP events."""
    out = {}
    ts, outcomes = schedule()
    for interval in (5, 8):
        upd = qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0, 1]), device_rng=True, seed=0)
        upd.batch_update(outcomes[:40], ts[:40], resample_interval=interval)      # untimed: allocator, first launches
        upd.reset()
        upd._resample_count = 0
        torch.cuda.synchronize()
        eng.set_profiling(1)
        upd.batch_update(outcomes, ts, resample_interval=interval)
        torch.cuda.synchronize()
        ms, tags = eng.profile_read()
        eng.set_profiling(0)
        upd.reset()
        rc0 = upd.resample_count
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        upd.batch_update(outcomes, ts, resample_interval=interval)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        kt = kernel_table(ms, tags)
        e = {"workload": "SimplePrecessionModel batch_update, %.0e particles, the %d data of the headline schedule, "
                         "resample_interval=%d" % (n, N_SCHEDULE, interval),
             "value": n * N_SCHEDULE / wall, "unit": "particle-updates/s", "ms_per_datum": wall / N_SCHEDULE * 1e3,
             "resamples": upd.resample_count - rc0, "posterior_mean": float(upd.est_mean()[0])}
        if "update_multi" in kt:
            # one window = one pass: read x (8 d) and w (8), write w (8), whatever the number of data in it
            e["window_kernel"] = frac_entry("k_update_multi<PRECESSION>", kt["update_multi"]["avg_us"], 24.0 * n,
                                            kt["update_multi"]["launches"],
                                            {"bytes_per_particle_per_window": 24, "data_per_window": interval if interval <= 8 else 8,
                                             "note": "VALU-bound: up to 8 likelihoods per particle per pass"})
        out["batch_update_interval_%d" % interval] = e
        del upd
        torch.cuda.empty_cache()
    # config 5's share through batch_update: 2-qubit tomography, random Pauli measurements, resample_interval 5 -- the window
    # kernel reads only the rows the window's measurement vectors touch (k_update_multi_tomo, round 5)
    try:
        spec5 = next(s5 for s5 in other_config_specs(qi) if s5["key"] == "config5_share_tomography")
        eps5 = np.concatenate(spec5["eps"])
        outs5 = np.asarray(spec5["outs"])
        n5, K5 = spec5["n"], len(outs5)
        upd = qi.SMCUpdater(spec5["model"], n5, spec5["prior"](), device_rng=True, seed=0)
        upd.batch_update(outs5, eps5, resample_interval=5)
        upd.reset()
        torch.cuda.synchronize()
        eng.set_profiling(1)
        upd.batch_update(outs5, eps5, resample_interval=5)
        torch.cuda.synchronize()
        ms, tags = eng.profile_read()
        eng.set_profiling(0)
        upd.reset()
        rc0 = upd.resample_count
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        upd.batch_update(outs5, eps5, resample_interval=5)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        kt = kernel_table(ms, tags)
        rows = float(np.mean([len({0} | {int(np.flatnonzero(eps5["meas"][k])[-1]) for k in range(i, min(i + 5, K5))})
                              for i in range(0, K5, 5)]))
        e = {"workload": "2-qubit TomographyModel batch_update, %.3g particles, the 60 random Pauli measurements of config 5, "
                         "resample_interval=5" % n5,
             "value": n5 * K5 / wall, "unit": "particle-updates/s", "ms_per_datum": wall / K5 * 1e3,
             "resamples": upd.resample_count - rc0, "posterior_mean_head": [float(v) for v in upd.est_mean()[:3]]}
        if "update_multi" in kt:
            e["window_kernel"] = frac_entry("k_update_multi_tomo<5,2>", kt["update_multi"]["avg_us"], (16.0 + 8.0 * rows) * n5,
                                            kt["update_multi"]["launches"],
                                            {"bytes_per_particle_per_window": 16.0 + 8.0 * rows, "rows_touched_per_window": rows,
                                             "data_per_window": 5,
                                             "note": "sparse measurement vectors: w in, w out and the union of the window's rows "
                                                     "(dense window: 144 B per particle)"})
        out["batch_update_tomography_interval_5"] = e
        del upd
        torch.cuda.empty_cache()
    except Exception as ex:  # noqa: BLE001
        out["batch_update_tomography_interval_5"] = {"error": repr(ex)}
    try:
        out["plugin_device_hook"] = plugin_paths(qi, eng, torch, n)
    except Exception as ex:  # noqa: BLE001
        out["plugin_device_hook"] = {"error": repr(ex)}
    m = qi.BinomialModel(qi.SimplePrecessionModel())
    upd = qi.SMCUpdater(m, n, qi.UniformDistribution([0, 1]), device_rng=True, seed=0)
    ep = np.empty((1,), dtype=m.expparams_dtype)
    ep['x'], ep['n_meas'] = 7.0, 25
    upd.update(11, ep)                                       # a non-trivial posterior to design against
    design = np.empty((4,), dtype=m.expparams_dtype)
    design['x'], design['n_meas'] = [3.0, 9.0, 14.0, 21.0], 25
    upd.bayes_risk(design[:1])
    torch.cuda.synchronize()
    eng.set_profiling(1)
    t0 = time.perf_counter()
    risk = upd.bayes_risk(design)
    wall = time.perf_counter() - t0
    ms, tags = eng.profile_read()
    eng.set_profiling(0)
    kt = kernel_table(ms, tags)
    e = {"workload": "bayes_risk of 4 hypothetical Binomial(SimplePrecession, n_meas=25) experiments (26 outcomes each), "
                     "%.0e particles" % n,
         "ms_per_experiment": wall / 4 * 1e3, "hypothetical_likelihoods_per_s": 26 * 4 * n / wall,
         "risk": [float(v) for v in risk]}
    if "hyp_sums" in kt:
        passes = kt["hyp_sums"]["launches"] // 4          # (launches per experiment: 26 outcomes = one two-ended pass of 13 + 13)
        e["kernel"] = frac_entry("k_hyp_sums_chain2<BINOMIAL_PRECESSION,MOMENTS,13>", kt["hyp_sums"]["avg_us"], 16.0 * n,
                                 kt["hyp_sums"]["launches"],
                                 {"bytes_per_particle": 16, "outcomes_per_pass": 26 // max(passes, 1), "passes_per_experiment": passes,
                                  "kernel_us_per_experiment": kt["hyp_sums"]["avg_us"] * passes,
                                  "note": "VALU-bound (~260 instructions per particle and pass, 59-69 % VALU-busy at four waves per "
                                          "SIMD): cos^2, two integer powers and one reciprocal per particle, then a geometric walk from "
                                          "both ends of the outcome range -- 1 multiply + 1 add + 2 multiply-adds per (particle, "
                                          "outcome) -- the two directions on partner waves (start values handed over through LDS), 39 "
                                          "running sums a lane; binomial coefficients applied on the host; the four experiments' "
                                          "passes queue back to back behind one wait"})
    out["bayes_risk_26_outcomes"] = e
    del upd
    torch.cuda.empty_cache()
    return out


def plugin_paths(qi, eng, torch, n=10_000_000, n_data=40):
    """The qinfer.Model plugin surface at the headline cloud size (abstract_model.py:444-468): UnknownT2Model
    (test_models.py:222-259; two parameters) served three ways through the same SMCUpdater.update loop --
      native        the library's own kernels (k_update_fused<UNKNOWN_T2> + the d = 2 sampler);
      hip_plugin    a user model WITHOUT native kernels that states its likelihood as HIP device source (`likelihood_hip`):
                    hiprtc compiles it into the fused update kernel (csrc/kernels/user_jit.hpp) -- one pass per datum;
      torch_plugin  a user model WITHOUT native kernels that defines `likelihood_device` / `are_models_valid_device`
                    (eager torch on the (d, N) tensor the cloud lives in: no host copy; the update is qsmc_update_from_
                    likelihood, the resample the Philox sampler + the model's own validity test);
      numpy_plugin  the same model with only the reference's NumPy methods (the cloud's host copy is kept between resamples;
                    its likelihood runs on the host: a few data only).
    ms per datum includes every resample the data trigger."""
    class NumpyT2(qi.FiniteOutcomeModel):
        n_modelparams = 2
        expparams_dtype = [('t', 'float')]
        is_n_outcomes_constant = True

        def n_outcomes(self, expparams):
            return 2

        def are_models_valid(self, modelparams):
            return np.all(modelparams >= 0, axis=1)

        def likelihood(self, outcomes, modelparams, expparams):
            super().likelihood(outcomes, modelparams, expparams)
            t = np.asarray(expparams['t'], dtype=float)[None, :]
            e = np.exp(-t * modelparams[:, 1:2])
            pr0 = e * np.cos(modelparams[:, 0:1] * t / 2) ** 2 + (1 - e) / 2
            return qi.FiniteOutcomeModel.pr0_to_likelihood_array(outcomes, pr0)

    class TorchT2(NumpyT2):
        def likelihood_device(self, outcomes, x_dev, expparams):
            self.count_likelihood_calls(len(outcomes), x_dev.shape[1], expparams.shape[0])
            rows = []
            for t in np.asarray(expparams['t'], dtype=float):
                e = torch.exp(x_dev[1] * (-float(t)))
                c = torch.cos(x_dev[0] * (float(t) / 2))
                c.mul_(c).sub_(0.5).mul_(e).add_(0.5)              # pr0 = e (cos^2 - 1/2) + 1/2, in place
                rows.append(c)
            pr0 = torch.stack(rows)
            return torch.stack([pr0 if int(o) == 0 else 1 - pr0 for o in outcomes])

        def are_models_valid_device(self, x_dev):
            return (x_dev >= 0).all(dim=0)
    class HipT2(NumpyT2):
        likelihood_hip = r"""
__device__ double likelihood(const double *x, const double *ep, long long outcome) {
    const double t = ep[0], e = exp(-t * x[1]), c = cos(x[0] * t / 2);
    const double pr0 = e * (c * c) + (1 - e) / 2;
    return outcome == 0 ? pr0 : 1 - pr0;
}
#define QSMC_USER_HAS_VALID 1
__device__ bool valid(const double *x) { return x[0] >= 0 && x[1] >= 0; }
"""
    rs = np.random.RandomState(0)
    ts = np.linspace(0.5, 14.0, n_data)
    e = np.exp(-ts * 0.05)
    outs = (rs.random_sample(n_data) >= e * np.cos(0.7 * ts / 2) ** 2 + (1 - e) / 2).astype(int)
    eps = np.array([(t,) for t in ts], dtype=[('t', 'float')])
    res = {"workload": "UnknownT2Model (omega, 1/T2), %.0e particles, %d data t = 0.5 .. 14, prior U[0, 1.5] x U[0, 0.2], "
                       "Liu-West a = 0.98, device RNG; SMCUpdater.update per datum" % (n, n_data)}
    for key, model, k_data in (("native", qi.UnknownT2Model(), n_data), ("hip_plugin", HipT2(), n_data),
                               ("torch_plugin", TorchT2(), n_data), ("numpy_plugin", NumpyT2(), 6)):
        upd = qi.SMCUpdater(model, n, qi.UniformDistribution([[0.0, 1.5], [0.0, 0.2]]), device_rng=True, seed=0)
        for k in range(min(k_data, 12)):                        # untimed: allocator, first launches (incl. one resample)
            upd.update(int(outs[k]), eps[k:k + 1])
        upd.resample()
        upd.reset()
        rc0 = upd.resample_count
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(k_data):
            upd.update(int(outs[k]), eps[k:k + 1])
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        res[key] = {"ms_per_datum": wall / k_data * 1e3, "value": n * k_data / wall, "unit": "particle-updates/s",
                    "data": k_data, "resamples": upd.resample_count - rc0,
                    "posterior_mean": [float(v) for v in upd.est_mean()]}
        del upd
        torch.cuda.empty_cache()
    # the same data through batch_update(resample_interval=5): fused windows for the native model AND for the compiled one
    for key, model in (("native", qi.UnknownT2Model()), ("hip_plugin", HipT2())):
        upd = qi.SMCUpdater(model, n, qi.UniformDistribution([[0.0, 1.5], [0.0, 0.2]]), device_rng=True, seed=0)
        upd.batch_update(outs[:15], eps[:15], resample_interval=5)
        for _ in range(3):                                      # untimed, as in the per-datum legs: a plugin resample's buffer
            upd.resample()                                      # with spares is allocated at its second call
            upd.update(int(outs[0]), eps[0:1], check_for_resample=False)
        upd.reset()
        rc0 = upd.resample_count
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        upd.batch_update(outs, eps, resample_interval=5)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        res[key]["batch_update_interval_5"] = {"ms_per_datum": wall / n_data * 1e3, "value": n * n_data / wall,
                                               "resamples": upd.resample_count - rc0}
        del upd
        torch.cuda.empty_cache()
    res["hip_plugin"]["vs_native"] = res["hip_plugin"]["ms_per_datum"] / res["native"]["ms_per_datum"]
    res["torch_plugin"]["vs_native"] = res["torch_plugin"]["ms_per_datum"] / res["native"]["ms_per_datum"]
    res["numpy_plugin"]["vs_native"] = res["numpy_plugin"]["ms_per_datum"] / res["native"]["ms_per_datum"]
    res["note"] = ("hip_plugin: the model's own device function inlined into one fused pass (32 B per particle), moments and "
                   "sums in the same pass; the resample runs on the Philox sampler with the model's compiled valid(); "
                   "torch_plugin: eight eager elementwise torch kernels per datum (each a pass over 80-240 MB) + the weight "
                   "update pass, against ONE fused pass of 32 B per particle for the native kernel")
    return res


def beyond_l3(qi, eng, torch, n=100_000_000, steps=12):
    """The update kernel on a cloud past the Infinity Cache: N = 1e8, 2.4 GB streamed per launch."""
    upd = qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0, 1]), device_rng=True, seed=0)
    for k in range(3):
        upd.update(k & 1, np.array([1.5 * (k + 1)]), check_for_resample=False)
    torch.cuda.synchronize()
    eng.set_profiling(1)
    t0 = time.perf_counter()
    for k in range(steps):
        upd.update(k & 1, np.array([2.0 + 0.37 * k]), check_for_resample=False)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms, tags = eng.profile_read()
    eng.set_profiling(0)
    full = ms[tags == 0]
    e = frac_entry("k_update_fused<PRECESSION,VEC=2,ONES=false>", float(full.mean()) * 1e3, 24.0 * n, int(len(full)),
                   {"particles": n, "working_set_bytes": 24 * n, "bound": "hbm",
                    "value_updates_only": n * steps / wall,
                    "note": "N = 1e8: x + two weight buffers = 2.4 GB, ten times the 256 MB Infinity Cache"})
    del upd
    torch.cuda.empty_cache()
    return e


# ------------------------------------------------------------------------------------------------ launcher
def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU, rendezvous
    on 127.0.0.1 and a free port.  Rank 0's JSON line passes through on stdout (the children inherit it)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["QSMC_BENCH_LAUNCHER"] = "self"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=float, default=1e7, help="particles PER GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--only", default=None, help="run ONE other config by key instead of the headline (profiling runs)")
    ap.add_argument("--cpu-data", type=int, default=20, help="data of the schedule the CPU baseline runs")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="put hipEvents on every N-th launch of each timed kernel kind (0 = steps // 5, 1..8)")
    ap.add_argument("--strong-particles", type=float, default=1e7,
                    help="particles IN TOTAL of the strong-scaling leg of a sharded run (BASELINE.json: 1e7 at 1/2/4/8 GPU)")
    ap.add_argument("--force-comm", action="store_true",
                    help="run the sharded code path even with one rank (validation on a 1-GPU box)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process becomes one (the ranks are its children under torch.distributed.run)
        if os.environ.get("QSMC_BENCH_SHARE_GPU") != "1":
            import torch
            have = torch.cuda.device_count()
            if have < args.gpus:
                raise SystemExit("--gpus %d but %d GPU(s) visible (QSMC_BENCH_SHARE_GPU=1 puts every rank on device 0: "
                                 "a control-flow check, not a measurement)" % (args.gpus, have))
        raise SystemExit(self_launch(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # validation hook for a 1-GPU box: QSMC_BENCH_SHARE_GPU=1 puts every rank on device 0 and talks over gloo, so
    # that the multi-rank control flow of this file can be exercised without N GPUs (not a measurement mode)
    share_gpu = os.environ.get("QSMC_BENCH_SHARE_GPU") == "1"
    if share_gpu:       # the resampler's redraw kernel needs all its workgroups resident: split the GPU's 512 slots
        os.environ.setdefault("QSMC_REDRAW_BLOCKS", str(max(16, 512 // max(world, 2))))
    device_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(device_index)
    comm = None
    if world > 1 or args.force_comm:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        from qinfer_amd.parallel import ParticleShardGroup
        comm = ParticleShardGroup()

    import qinfer_amd as qi
    from qinfer_amd.engine import get_engine
    eng = get_engine()
    n = int(args.particles)
    ts, outcomes = schedule()
    stride = args.event_stride if args.event_stride > 0 else max(1, min(8, args.steps // 5))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def reduce_max(w_):
        """The slowest rank's wall time."""
        if world == 1:
            return float(w_)
        wt_ = torch.tensor([w_], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        torch.distributed.all_reduce(wt_, op=torch.distributed.ReduceOp.MAX)
        return float(wt_.item())

    if args.only == "plugin_paths":    # (the plugin surface alone)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            print(json.dumps({"plugin_device_hook": plugin_paths(qi, eng, torch)}), flush=True)
        return
    if args.only == "other_paths":     # (SURVEY 8(f) rows alone: profiling runs)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            print(json.dumps({"other_paths": other_paths(qi, eng, torch)}), flush=True)
        return
    if args.only and args.only not in ("shard_preview",):   # one of the other configs alone (what the per-config rocprofv3 passes run)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            spec = next(s for s in other_config_specs(qi) if s["key"] == args.only)
            print(json.dumps({args.only: run_other_config(qi, eng, torch, spec, args.warmup)}), flush=True)
        return

    def k_steps(upd):
        for i in range(args.steps):
            k = i % N_SCHEDULE
            if i and k == 0:
                upd.reset()
            upd.update(int(outcomes[k]), ts[k:k + 1])

    def timed_pass(upd, events, repeats=0):
        """W warm-up data of a throwaway pass, reset, then exactly --steps updates between barriers.  Returns the wall time
        of that pass and of `repeats` further identical passes (reported beside it, never as `value`).
        Untimed, besides the W data: one forced resample + one update (with --warmup 5 the warm-up data trigger no
        resample: the first one of the process -- scratch growth, first launches of six kernels -- would sit inside a
        3 ms timed region; round 3's driver line read 0.177 ms/step against 0.077 for the same sample a minute later) and
        one rehearsal of the K steps themselves.  The garbage collector is off inside the timed region, as in `timeit`."""
        import gc
        # (collected once, up front, then off until the last pass is done: a full collection right before the timed region
        #  is tens of milliseconds of idle GPU -- measured: the passes behind it ran 0.085-0.096 ms/step, clocks ramping
        #  back up, against 0.077 for the same 20 data in a busy process)
        gc.collect()
        gc.disable()
        try:                                                    # (a failure in here must not leave the collector off)
            for k in range(args.warmup):
                upd.update(int(outcomes[k % N_SCHEDULE]), ts[k % N_SCHEDULE:k % N_SCHEDULE + 1])
            upd.resample()
            upd.update(int(outcomes[0]), ts[0:1])
            upd.reset()
            k_steps(upd)
            walls = []
            for rep in range(1 + repeats):
                upd.reset()
                upd._resample_count = 0
                # a launch that carries start/stop events drains the queue around itself (measured: 10.7 us per step
                # with every launch timed), so every `stride`-th launch of each kernel kind is timed
                eng.set_profiling(stride if (events and rep == 0) else 0)
                # (in the timed region only the dominant kernel carries events -- the update, tags 0 and 2: what `roofline`
                #  is made of; the sampler / counts figures come from the census right after, every launch timed)
                eng.set_profiling_tags((0, 2) if (events and rep == 0 and world == 1 and comm is None) else None)
                barrier()
                t0 = time.perf_counter()
                k_steps(upd)
                barrier()
                walls.append(time.perf_counter() - t0)
                if rep == 0:
                    # (est_mean is a collective on a sharded cloud: every rank is here)
                    first_pass.append((upd.resample_count,) + tuple(eng.profile_read() if events else (np.zeros(0), np.zeros(0)))
                                      + (float(upd.est_mean()[0]),))
        finally:
            gc.enable()
            eng.set_profiling_tags(None)
        return walls[0], walls[1:]

    first_pass = []        # (resamples, kernel durations [ms], kernel tags, posterior mean) of each call's contract pass

    def batch_pass(upd, interval):
        """The same --steps data through `batch_update(resample_interval=interval)` (smc.py:459-487): the data between two
        n_ess tests go through ONE pass over the cloud (k_update_multi) and -- on a sharded cloud -- ONE reduction per
        window instead of one per datum.  One untimed rehearsal, reset, then the timed call between barriers.  Returns
        (wall seconds of this rank, resamples, posterior mean)."""
        import gc
        idx = np.arange(args.steps) % N_SCHEDULE
        chunks = [(outcomes[idx[i:i + N_SCHEDULE]], ts[idx[i:i + N_SCHEDULE]]) for i in range(0, args.steps, N_SCHEDULE)]

        def go():
            for j, (oc, tt) in enumerate(chunks):
                if j:
                    upd.reset()
                upd.batch_update(oc, tt, resample_interval=interval)
        gc.collect()
        gc.disable()
        try:
            upd.reset()
            go()
            upd.reset()
            upd._resample_count = 0
            barrier()
            t0 = time.perf_counter()
            go()
            barrier()
            wall_b = time.perf_counter() - t0
        finally:
            gc.enable()
        return wall_b, upd.resample_count, float(upd.est_mean()[0])

    def batch_legs(upd, n_total_particles, reduce_max):
        """`batch_update_interval_{5,8}` entries for an updater already warmed by a timed_pass."""
        out = {}
        for interval in (5, 8):
            try:
                wall_b, res_b, mean_b = batch_pass(upd, interval)
                wall_b = reduce_max(wall_b)
                out["batch_update_interval_%d" % interval] = {
                    "value": n_total_particles * args.steps / wall_b, "unit": "particle-updates/s",
                    "ms_per_datum": wall_b / args.steps * 1e3, "resamples": res_b, "posterior_mean": mean_b,
                    "data": args.steps, "windows": -(-args.steps // interval)}
            except Exception as e:  # noqa: BLE001
                out["batch_update_interval_%d" % interval] = {"error": repr(e)}
        return out

    def other_transport_pass(transport):
        """The same K steps once more under the OTHER transport of the per-datum reduction (`value` is the pass with
        whatever ParticleShardGroup picked: `auto` measures the two at group creation when every rank has a GPU of its
        own, else host shared memory on one node): "rccl" = the library's own RCCL collective on the launch stream
        (north_star's transport), "shm" = host shared memory.  Returns the `transports[transport]` entry; under "rccl"
        `ranks_in_comm` is what the communicator itself reports (ncclCommCount)."""
        try:
            from qinfer_amd.parallel import ParticleShardGroup
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                comm_r = ParticleShardGroup(transport=transport)
                upd_r = qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0, 1]),
                                      device_rng=True, seed=0, comm=comm_r)
                wall_r, _ = timed_pass(upd_r, events=False)
                resamples_r, _, _, mean_r = first_pass[-1]
                wall_r = reduce_max(wall_r)
                res = {"per_datum_collective": comm_r.transport_name,
                       "value": n * world * args.steps / wall_r,
                       "ms_per_step": wall_r / args.steps * 1e3, "resamples": resamples_r,
                       "posterior_mean": mean_r}
                if transport == "rccl":
                    res["ranks_in_comm"] = comm_r.ranks_in_comm(eng)[0]
                comm_r.close()
        except Exception as e:  # noqa: BLE001
            res = {"error": repr(e)}
        return res

    def shard_preview(n_shard):
        """The headline workload on ONE rank's share of a strong-scaling run (1e7 particles over 8 GPUs = 1.25e6 each):
        what a GPU of the 8-GPU point does between two collectives, measured where it can be -- on one GPU, no
        communication.  At this size a datum is launch / round-trip bound, not bandwidth bound."""
        upd_p = qi.SMCUpdater(qi.SimplePrecessionModel(), n_shard, qi.UniformDistribution([0, 1]), device_rng=True, seed=0)
        wall_p, _ = timed_pass(upd_p, events=False)
        res_p, _, _, mean_p = first_pass[-1]
        upd_p.reset()
        eng.set_profiling(1)
        for k in range(min(64, N_SCHEDULE)):
            upd_p.update(int(outcomes[k]), ts[k:k + 1])
        torch.cuda.synchronize()
        p_ms, p_tags = eng.profile_read()
        eng.set_profiling(False)
        kt = kernel_table(p_ms, p_tags)
        out = {"workload": "SimplePrecessionModel SMCUpdater.update, %.3g particles (one rank's share of 1e7 over 8 GPUs), "
                           "same schedule and resampler as the headline, one GPU, no collective" % n_shard,
               "particles": n_shard, "steps": args.steps, "value": n_shard * args.steps / wall_p,
               "ms_per_step": wall_p / args.steps * 1e3, "resamples": res_p, "posterior_mean": mean_p,
               "eight_of_these_without_a_collective": 8 * n_shard * args.steps / wall_p}
        if "update" in kt:
            out["update_kernel"] = frac_entry("k_update_fused<PRECESSION,VEC=2,ONES=false>", kt["update"]["avg_us"],
                                              24.0 * n_shard, kt["update"]["launches"], {"bytes_per_particle": 24})
        if "sample" in kt:
            out["resample_kernel"] = frac_entry("k_bucket_sample<D=1,512>", kt["sample"]["avg_us"], 24.0 * n_shard,
                                                kt["sample"]["launches"], {"bytes_per_particle": 24})
        # the same data through batch_update: the path that amortises the per-datum fixed cost (launch + reduction + host
        # round trip) over a window -- what bounds a shard of this size under update()
        out.update(batch_legs(upd_p, n_shard, lambda w_: w_))
        for k_ in ("batch_update_interval_5", "batch_update_interval_8"):
            if "value" in out.get(k_, {}):
                out[k_]["vs_update"] = out[k_]["value"] / out["value"]
        del upd_p
        torch.cuda.empty_cache()
        return out

    if args.only == "shard_preview":     # (the strong-scaling shard preview alone: tests / profiling runs)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            print(json.dumps({"strong_scaling_shard_preview": shard_preview(max(4096, int(args.strong_particles) // 8))}),
                  flush=True)
        return

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        upd = qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0, 1]),
                            device_rng=True, seed=0, comm=comm)
        wall, repeat_walls = timed_pass(upd, events=not os.environ.get("QSMC_BENCH_NO_EVENTS"), repeats=2)
        # the kernels' start/stop events (hipExtLaunchKernelGGL, on the launch stream: those of the first, the contract's,
        # pass) are read here, once
        eng.set_profiling(False)
        resamples_timed, all_ms, tags, posterior_mean = first_pass[0]
        # C2's own schedule (k = 0..199, 70 resamples) whatever --steps says: the driver's 20-step command sees
        # 3 resamples in 20 data (15 %), the full schedule 35 % -- both figures in the line
        headline_200 = None
        if args.steps != N_SCHEDULE and not os.environ.get("QSMC_BENCH_NO_EVENTS"):
            try:
                import gc
                gc.collect()
                gc.disable()
                try:
                    def pass_200():
                        upd.reset()
                        upd._resample_count = 0
                        barrier()
                        t0 = time.perf_counter()
                        for k in range(N_SCHEDULE):
                            upd.update(int(outcomes[k]), ts[k:k + 1])
                        barrier()
                        return time.perf_counter() - t0
                    pass_200()                               # (rehearsal: later resamples' first launches, allocator)
                    w200 = pass_200()
                finally:
                    gc.enable()
                w200_t = torch.tensor([w200], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
                if world > 1:
                    torch.distributed.all_reduce(w200_t, op=torch.distributed.ReduceOp.MAX)
                w200 = float(w200_t.item())
                headline_200 = {"value": n * world * N_SCHEDULE / w200, "unit": "particle-updates/s",
                                "ms_per_step": w200 / N_SCHEDULE * 1e3, "steps": N_SCHEDULE,
                                "resamples": upd.resample_count, "posterior_mean": float(upd.est_mean()[0]),
                                "what": "the headline workload over its whole schedule (t_k = (9/8)^k, k = 0..199), same "
                                        "updater, right after the contract's timed region; no kernel events"}
                # The same two figures on the REFERENCE's outcome sequence (SURVEY 8(d) defines C2's data as G1's: the
                # outcomes the reference simulated under np.random.seed(0), kept in tests/golden).  The headline's own
                # sequence -- RandomState(0) draws from the same model on the same schedule, unchanged since round 1 so
                # that rounds compare -- happens to drive the posterior onto an alias (0.30012; the reference's
                # algorithm does the same on it: tests/test_gpu_parity.py::test_device_rng_trajectory_statistics_vs_oracle)
                # and resamples 70 times in 200 data, 3 in the first 20; G1's resamples ~43 times, 5 in the first 20.
                try:
                    g1 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden",
                                              "g1_precession_n1000.npz"))
                    g1_out = g1["outcomes"].astype(np.int64)
                    gc.collect()
                    gc.disable()
                    try:
                        def pass_g1(n_data):
                            upd.reset()
                            upd._resample_count = 0
                            barrier()
                            t0 = time.perf_counter()
                            for k in range(n_data):
                                upd.update(int(g1_out[k]), ts[k:k + 1])
                            barrier()
                            return time.perf_counter() - t0
                        pass_g1(N_SCHEDULE)
                        entry = {}
                        for n_data in (N_SCHEDULE, min(args.steps, N_SCHEDULE)):
                            w_g1 = reduce_max(pass_g1(n_data))
                            entry["steps_%d" % n_data] = {"value": n * world * n_data / w_g1, "ms_per_step": w_g1 / n_data * 1e3,
                                                          "resamples": upd.resample_count,
                                                          "posterior_mean": float(upd.est_mean()[0])}
                        entry["what"] = ("the headline workload on the reference's own simulated outcomes (fixture "
                                         "g1_precession_n1000: SURVEY 8(d)'s data for config 2) instead of the bench's "
                                         "RandomState(0) draws; same updater, same schedule")
                        headline_200["reference_outcome_sequence"] = entry
                    finally:
                        gc.enable()
                except Exception as e:  # noqa: BLE001
                    headline_200["reference_outcome_sequence"] = {"error": repr(e)}
            except Exception as e:  # noqa: BLE001
                headline_200 = {"error": repr(e)}
        # tag 0: update with explicit weights (24 B/particle), 2: first update after a reset/resample, weights
        # implicit (16 B/particle), 1: the resampler's sampling kernel, 6: its counts/plan launch
        full_ms, ones_ms, sampler_ms = all_ms[tags == 0], all_ms[tags == 2], all_ms[tags == 1]

        # census (untimed for `value`): the first 64 data of the schedule once more, EVERY launch timed
        census = None
        gpu_same_sample = None
        if world == 1 and comm is None:
            upd.reset()
            eng.set_profiling(1)
            for k in range(64):
                upd.update(int(outcomes[k]), ts[k:k + 1])
            torch.cuda.synchronize()
            c_ms, c_tags = eng.profile_read()
            eng.set_profiling(False)
            census = kernel_table(c_ms, c_tags)
            # the CPU baseline's sample (first --cpu-data data) on the GPU, no events: like against like
            upd.reset()
            rc0 = upd.resample_count
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(args.cpu_data):
                upd.update(int(outcomes[k]), ts[k:k + 1])
            torch.cuda.synchronize()
            w1 = time.perf_counter() - t1
            gpu_same_sample = {"value": n * args.cpu_data / w1, "ms_per_step": w1 / args.cpu_data * 1e3,
                               "resamples": upd.resample_count - rc0}

    wall_t = torch.tensor([wall], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
    if world > 1:
        torch.distributed.all_reduce(wall_t, op=torch.distributed.ReduceOp.MAX)
    wall = float(wall_t.item())
    del upd
    torch.cuda.empty_cache()

    extras = {}
    if rank == 0 and world == 1 and comm is None and not args.no_other_configs:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                extras["roofline_beyond_l3"] = beyond_l3(qi, eng, torch)
            except Exception as e:  # noqa: BLE001
                extras["roofline_beyond_l3"] = {"error": repr(e)}
            try:
                extras["strong_scaling_shard_preview"] = shard_preview(max(4096, int(args.strong_particles) // 8))
            except Exception as e:  # noqa: BLE001
                extras["strong_scaling_shard_preview"] = {"error": repr(e)}
            oc = {}
            for spec in other_config_specs(qi):
                if spec.get("extra"):
                    continue
                try:
                    oc[spec["key"]] = run_other_config(qi, eng, torch, spec, min(args.warmup, 5))
                except Exception as e:  # noqa: BLE001
                    oc[spec["key"]] = {"error": repr(e)}
            extras["other_configs"] = oc
            try:
                extras["other_paths"] = other_paths(qi, eng, torch)
            except Exception as e:  # noqa: BLE001
                extras["other_paths"] = {"error": repr(e)}

    def sharded_configs(comm_s):
        """BASELINE configs 4 and 5 as they are defined -- a cloud sharded over the ranks (1e8 RB particles = 1.25e7 per
        rank at 8 ranks; 1e7 two-qubit tomography particles = 1.25e6 per rank) -- through `comm_s`: every rank runs its
        shard, the per-datum sums and the resample's moments cross ranks, value = all ranks' particle-updates / the slowest
        rank's wall time.  QSMC_BENCH_SHARE_GPU=1 (every rank on device 0) shrinks the shards: control flow, not a
        measurement."""
        out = {}
        for spec in other_config_specs(qi):
            if spec["key"] not in ("config4_share_rb", "config5_share_tomography"):
                continue
            key = spec["key"].replace("_share", "_sharded")
            n_rank = None
            if share_gpu or os.environ.get("QSMC_BENCH_SHARDED_PARTICLES"):
                n_rank = int(float(os.environ.get("QSMC_BENCH_SHARDED_PARTICLES", "0"))) or max(65536, spec["n"] // (16 * world))
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    out[key] = run_other_config(qi, eng, torch, spec, min(args.warmup, 5), comm=comm_s,
                                                n_override=n_rank, sync=barrier)
            except Exception as e:  # noqa: BLE001
                out[key] = {"error": repr(e)}
        return out

    def drain_c_stdio():
        # RCCL prints a version banner through C stdio, which (not a tty) would be flushed at exit -- after the
        # JSON line.  Push whatever C stdio holds to stderr so that the JSON line is the last line of stdout.
        import ctypes
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
    drain_c_stdio()

    if os.environ.get("QSMC_BENCH_NO_EVENTS"):
        # diagnostic only (not the bench line): the same loop without the per-kernel events
        if rank == 0:
            print(json.dumps({"diagnostic": "no kernel events", "ms_per_step": wall / args.steps * 1e3}), flush=True)
        return
    if rank == 0:
        n_total = n * world
        traffic = load_traffic()
        # (under local placement a rank's shard size floats by ~1e-3 relative; n is exact at N = 1)
        full_bytes, ones_bytes = float(24 * n), float(16 * n)
        if len(full_ms) == 0 and census and "update" in census:     # (--steps too small for one sampled launch)
            full_ms = np.array([census["update"]["avg_us"] * 1e-3])
        avg_kernel_s = float(full_ms.mean()) * 1e-3          # the dominant variant: reads x and w, writes w
        achieved = full_bytes / avg_kernel_s / 1e9
        ones_info = None
        if len(ones_ms):
            ones_s = float(ones_ms.mean()) * 1e-3
            ones_info = {"timed_launches": int(len(ones_ms)), "algorithmic_bytes_per_launch": ones_bytes,
                         "avg_kernel_us": ones_s * 1e6, "achieved": ones_bytes / ones_s / 1e9}
        line = {
            "metric": "particle-updates/sec", "value": n_total * args.steps / wall,
            "unit": "particle-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "SimplePrecessionModel SMCUpdater.update, %.0e particles/GPU, fp64, "
                                   "Liu-West a=0.98, t_k=(9/8)^k" % n,
                       "particles_per_gpu": n, "particles_total": n_total,
                       "resamples_in_timed_region": resamples_timed, "rng": "philox4x32-10 (device)",
                       "parallelism": "particle-shard x%d" % world,
                       "per_datum_collective": (None if comm is None else comm.transport_name),
                       "rebalances_in_timed_region": (None if comm is None else comm.n_rebalances)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic.get("update_kernel_bytes_per_launch"),
                         "traffic_source": "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                           "command, committed; not collected by this run)",
                         "kernel": "k_update_fused<PRECESSION,VEC=2,ONES=false>", "avg_kernel_us": avg_kernel_s * 1e6,
                         "timed_launches": int(len(full_ms)), "event_stride": stride,
                         "algorithmic_bytes_per_launch": full_bytes,
                         "working_set_note": "240 MB at N = 1e7 fits the 256 MB Infinity Cache: see roofline_beyond_l3 for "
                                             "the HBM-only figure",
                         "implicit_uniform_weight_variant": ones_info},
            "posterior_mean": posterior_mean,
            "headline_200_steps": headline_200,
            # the same K steps twice more right after the contract's pass (no kernel events): a record of how far one 3 ms
            # region can sit from the next on this box -- never used for `value`
            "repeat_passes_ms_per_step": [w / args.steps * 1e3 for w in repeat_walls],
        }
        samp_n, samp_us = 0, None
        if len(sampler_ms):
            samp_n, samp_us = int(len(sampler_ms)), float(sampler_ms.mean()) * 1e3
        if census and "sample" in census and census["sample"]["launches"] > samp_n:
            samp_n, samp_us = census["sample"]["launches"], census["sample"]["avg_us"]
            samp_src = "census"
        else:
            samp_src = "timed region"
        if samp_us:
            # the resampler's main kernel, same clock: reads w (8 B), gathers x (8d), writes x' (8d) per particle
            line["resample_kernel"] = frac_entry("k_bucket_sample<D=1,512>", samp_us, float((8 + 16) * n), samp_n,
                                                 {"source": samp_src,
                                                  "traffic": traffic.get("sample_kernel_bytes_per_launch")})
        if census:
            line["kernel_census"] = {"what": "first 64 data of the schedule replayed right after the timed region, every "
                                             "launch timed (HIP events on the launch stream)", "kernels": census}
            if "update" in census:
                line["roofline"]["census_avg_kernel_us"] = census["update"]["avg_us"]
                line["roofline"]["census_launches"] = census["update"]["launches"]
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(n, args.cpu_data, gpu_same_sample)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(e)}
        line["config"]["launcher"] = os.environ.get("QSMC_BENCH_LAUNCHER", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ
                                                    else "none")
    else:
        line = None

    # ---- the transports of the per-datum reduction, both inside the line (sharded runs only)
    import threading
    done = {"printed": False}
    line_lock = threading.RLock()      # the main thread, the watchdog timer and the SIGTERM watcher all touch `line`

    def emit():
        # `printed` turns true only AFTER the line has been flushed, and the whole thing sits under the lock: a side
        # thread (watchdog / SIGTERM watcher) that calls os._exit right after its own emit() either finds the line already
        # out, or prints it itself -- never exits between a set flag and an unflushed line, never dumps a half-edited dict
        with line_lock:
            if rank == 0 and not done["printed"]:
                drain_c_stdio()
                print(json.dumps(line), flush=True)
                done["printed"] = True

    def set_line(path, value):
        """line[path[0]][path[1]]... = value, under the lock (rank 0 only holds a line)."""
        if rank != 0:
            return
        with line_lock:
            if done["printed"]:
                return
            node = line
            for k_ in path[:-1]:
                node = node[k_]
            node[path[-1]] = value

    watchdog = None
    if comm is not None:
        key = "rccl" if comm.transport == "rccl" else ("shm" if comm.transport_name == "host shared memory" else "backend")
        set_line(("transports",), {key: {"per_datum_collective": comm.transport_name, "value": None if line is None else line["value"],
                                         "ms_per_step": None if line is None else line["ms_per_step"],
                                         "resamples": resamples_timed, "posterior_mean": posterior_mean, "headline": True}})
        set_line(("config", "headline_transport"), key)       # which transport `value` was measured under
        # transport="auto" measures both transports at group creation when every rank has a GPU of its own
        # (ParticleShardGroup._probe_transports): both timings and the choice, or why nothing was measured
        set_line(("config", "transport_probe"), comm.transport_probe if comm.transport_probe is not None else {
            "skipped": "not measured: %s" % ("QSMC_BENCH_SHARE_GPU=1 (one device, gloo)" if share_gpu else
                                             "one rank" if world == 1 else "transport forced or no shared host")})
        want_sharded = not args.no_other_configs
        want_strong = (world > 1 or args.force_comm) and not os.environ.get("QSMC_BENCH_NO_STRONG")
        # the transport `value` was NOT measured under gets a pass of its own ("rccl" unless auto / QSMC_TRANSPORT picked it)
        other = "shm" if key == "rccl" else "rccl"
        want_rccl = ((world > 1 or os.environ.get("QSMC_BENCH_FORCE_RCCL_PASS"))
                     and not os.environ.get("QSMC_BENCH_NO_RCCL_PASS"))
        # everything after the headline (the strong-scaling leg, the sharded configs 4 and 5, the RCCL pass) runs under a
        # watchdog: a rank that hangs or a collective that never returns must not take the line with it.  Past the
        # deadline rank 0 prints the line with the time-out recorded against the stage that was running, and every rank
        # leaves.
        deadline = float(os.environ.get("QSMC_BENCH_DEADLINE", os.environ.get(
            "QSMC_BENCH_RCCL_DEADLINE", "420" if want_sharded else "90")))
        stage = {"name": "strong_scaling" if want_strong else ("sharded_configs" if want_sharded else "rccl")}

        def mark_stage(msg):
            """Record `msg` against the stage that was running (rank 0, under the line's lock)."""
            if rank != 0:
                return
            with line_lock:
                if done["printed"]:
                    return
                name = stage["name"]
                if name == "strong_scaling":
                    line["strong_scaling"] = msg
                elif name == "strong_scaling_rccl":
                    line["strong_scaling"].setdefault("transports", {})[other] = msg
                elif name == "sharded_configs":
                    line["sharded_configs"] = msg
                elif name == "sharded_configs_rccl":
                    line["sharded_configs"]["rccl_transport"] = msg
                else:
                    line["transports"][other] = msg

        def give_up():
            mark_stage({"error": "no result within %.0f s of the headline (watchdog), stage: %s" % (deadline, stage["name"])})
            emit()
            os._exit(0)
        if want_sharded or want_strong or (want_rccl and not share_gpu):
            watchdog = threading.Timer(deadline, give_up)
            watchdog.daemon = True
            watchdog.start()
        if world > 1:
            # a rank that DIES in one of these stages makes the launcher SIGTERM the others: rank 0 still prints the
            # line (headline intact, the stage marked) before it goes.  The C-level handler writes to a pipe whichever
            # thread takes the signal; the watcher thread runs even while the main thread sits in a C call.
            import signal
            rfd, wfd = os.pipe()
            os.set_blocking(wfd, False)
            signal.signal(signal.SIGTERM, lambda *a: None)
            signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)

            def on_sigterm():
                os.read(rfd, 1)
                mark_stage({"error": "terminated by the launcher (another rank failed), stage: %s" % stage["name"]})
                emit()
                os._exit(1)
            threading.Thread(target=on_sigterm, daemon=True).start()
            if os.environ.get("QSMC_BENCH_TEST_DIE_RANK") == str(rank):      # (tests/test_bench_launch.py only)
                os.kill(os.getpid(), signal.SIGKILL)

        # ---- strong scaling: BASELINE.json quotes its metric on "1e7 particles ... at 1/2/4/8 GPU" -- 1e7 in TOTAL.  The
        # headline above is the weak-scaling reading (1e7 per GPU); this leg holds the total fixed: every rank gets
        # 1e7 / world particles, same model, schedule and resampler, the per-datum reduction under each transport.
        strong_total = int(float(os.environ.get("QSMC_BENCH_STRONG_PARTICLES", "0")) or args.strong_particles)
        if share_gpu and not os.environ.get("QSMC_BENCH_STRONG_PARTICLES"):
            strong_total = max(65536 * world, strong_total // 16)          # (control flow on one shared device)
        n_strong = max(1, strong_total // world)

        def strong_leg(transport):
            """SimplePrecession, n_strong particles per rank (strong_total over all ranks), the headline's K steps."""
            from qinfer_amd.parallel import ParticleShardGroup
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    grp = ParticleShardGroup(transport=transport)
                    upd_s = qi.SMCUpdater(qi.SimplePrecessionModel(), n_strong, qi.UniformDistribution([0, 1]),
                                          device_rng=True, seed=0, comm=grp)
                    rb0 = grp.n_rebalances
                    wall_s, _ = timed_pass(upd_s, events=False)
                    resamples_s, _, _, mean_s = first_pass[-1]
                    wall_s = reduce_max(wall_s)
                    res = {"per_datum_collective": grp.transport_name, "value": n_strong * world * args.steps / wall_s,
                           "ms_per_step": wall_s / args.steps * 1e3, "resamples": resamples_s,
                           "rebalances": grp.n_rebalances - rb0, "posterior_mean": mean_s}
                    if grp.transport == "rccl":
                        res["ranks_in_comm"] = grp.ranks_in_comm(eng)[0]
                    # the same data through batch_update windows: one pass over the shard and ONE reduction per window --
                    # what amortises the per-datum fixed cost that bounds a 1e7 / world shard under update()
                    res.update(batch_legs(upd_s, n_strong * world, reduce_max))
                    for k_ in ("batch_update_interval_5", "batch_update_interval_8"):
                        if "value" in res.get(k_, {}):
                            res[k_]["vs_update"] = res[k_]["value"] / res["value"]
                    del upd_s
                    grp.close()
            except Exception as e:  # noqa: BLE001
                res = {"error": repr(e)}
            return res

        if want_strong:
            leg = strong_leg(key if key in ("shm", "rccl") else None)      # (the headline's transport, not a second probe)
            set_line(("strong_scaling",), {
                "scaling": "strong", "metric": "particle-updates/sec",
                "workload": "SimplePrecessionModel SMCUpdater.update, %.3g particles IN TOTAL over %d rank(s) (%.3g per "
                            "rank), fp64, Liu-West a=0.98, t_k=(9/8)^k: BASELINE.json's '1e7 particles at 1/2/4/8 GPU' "
                            "read as a fixed total" % (n_strong * world, world, n_strong),
                "particles_total": n_strong * world, "particles_per_rank": n_strong, "ranks": world, "steps": args.steps,
                "warmup": args.warmup, "value": leg.get("value"), "ms_per_step": leg.get("ms_per_step"),
                "value_transport": key,
                "batch_update_interval_5": leg.get("batch_update_interval_5"),
                "batch_update_interval_8": leg.get("batch_update_interval_8"),
                "transports": {key: leg}})
        sharded = None
        if want_sharded:
            stage["name"] = "sharded_configs"
            sharded = sharded_configs(comm)
            set_line(("sharded_configs",), sharded)
        stage["name"] = "rccl"
        if want_rccl and share_gpu:
            skipped = {"skipped": "QSMC_BENCH_SHARE_GPU=1: every rank sits on device 0 (control-flow check); an RCCL "
                                  "communicator needs one GPU per rank"}
            set_line(("transports", "rccl"), skipped)
            if want_strong and rank == 0 and isinstance(line.get("strong_scaling"), dict) and "transports" in line["strong_scaling"]:
                set_line(("strong_scaling", "transports", "rccl"), skipped)
        elif want_rccl:
            res = other_transport_pass(other)
            set_line(("transports", other), res)
            if want_strong and "error" not in res:
                stage["name"] = "strong_scaling_rccl"
                leg_r = strong_leg(other)
                if rank == 0 and isinstance(line.get("strong_scaling"), dict) and "transports" in line["strong_scaling"]:
                    set_line(("strong_scaling", "transports", other), leg_r)
            if sharded is not None and "error" not in res and other == "rccl":
                # configs 4 and 5 once more with the RCCL collective carrying the per-datum reduction
                stage["name"] = "sharded_configs_rccl"
                try:
                    from qinfer_amd.parallel import ParticleShardGroup
                    comm_r = ParticleShardGroup(transport="rccl")
                    sh_r = sharded_configs(comm_r)
                    comm_r.close()
                except Exception as e:  # noqa: BLE001
                    sh_r = {"error": repr(e)}
                if rank == 0:
                    with line_lock:
                        for k2, v2 in sh_r.items():
                            if isinstance(line["sharded_configs"].get(k2), dict) and isinstance(v2, dict):
                                line["sharded_configs"][k2]["rccl_transport"] = {
                                    kk: v2.get(kk) for kk in ("value", "ms_per_step", "resamples", "rebalances",
                                                              "per_datum_collective", "error") if kk in v2}
        if watchdog is not None:
            watchdog.cancel()
    emit()
    if world > 1 or args.force_comm:
        sys.stdout.flush()
        try:
            comm.close()
            torch.distributed.destroy_process_group()
        finally:
            if watchdog is not None:
                os._exit(0)                               # (RCCL's own threads must not keep the process alive)


if __name__ == "__main__":
    main()
