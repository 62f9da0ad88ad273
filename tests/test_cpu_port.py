"""Pins the C / OpenMP restatement (oracle/cpu_port.c -- the CPU baseline bench.py times) against the reference's
golden vectors: likelihood KATs (G2), the legacy MT19937 stream (NumPy itself), and the seeded trajectories of the
reference (G1 / G8) replayed from their recorded draws, datum by datum, with the tolerances of parity_tols.py.
No GPU, no product code."""
import warnings

import numpy as np
import pytest

import cpu_port as cp
import np_oracle as orc
import parity_tols as tol


def test_mt19937_stream_is_numpys():
    for seed in (0, 1, 12345, 2 ** 32 - 1):
        np.random.seed(seed)
        u = np.random.random(1500)
        np.testing.assert_array_equal(cp.mt_random(seed, 1500), u)
        np.random.seed(seed)
        z = np.random.randn(1501)                 # odd count: the cached second value of the polar method is used
        np.testing.assert_array_equal(cp.mt_randn(seed, 1501), z)


def test_g2_likelihoods(golden):
    g = golden("g2_likelihoods")
    x, ts = g["prec_x"], g["prec_t"]
    for o in (0, 1):
        for e, t in enumerate(ts):
            L = cp.likelihood(cp.PRECESSION, x, o, t=t)
            np.testing.assert_allclose(L, g["prec_L"][o, :, e], rtol=0, atol=1e-15)     # libm cos vs NumPy's: <= 4 ulp(1)
    x, ts, ns = g["bin_x"], g["bin_t"], g["bin_n"]
    for k in range(26):
        for e in range(len(ts)):
            L = cp.likelihood(cp.BINOMIAL_PRECESSION, x, k, t=ts[e], n_meas=int(ns[e]))
            np.testing.assert_allclose(L, g["bin_L"][k, :, e], rtol=1e-12, atol=1e-300)
    x, ms = g["rb_x"], g["rb_m"]
    for o in (0, 1):
        for e, m in enumerate(ms):
            L = cp.likelihood(cp.RB, x, o, m=int(m))
            np.testing.assert_allclose(L, g["rb_L"][o, :, e], rtol=0, atol=1e-15)
    x, meas = g["tomo_x"], g["tomo_meas"]
    for o in (0, 1):
        for e in range(meas.shape[0]):
            L = cp.likelihood(cp.TOMOGRAPHY, x, o, meas=meas[e])
            np.testing.assert_allclose(L, g["tomo_L"][o, :, e], rtol=0, atol=4e-16)


def test_g2_edges(golden):
    g = golden("g2_edges")
    x, ts = g["prec_x"], g["prec_t"]
    for o in (0, 1):
        for e, t in enumerate(ts):
            L = cp.likelihood(cp.PRECESSION, x, o, t=t)
            np.testing.assert_allclose(L, g["prec_L"][o, :, e], rtol=0, atol=1e-15)     # NaN matches NaN
    ks, ns = g["bin_k"], g["bin_n"]
    for e in range(len(ns)):
        for i, k in enumerate(ks):
            if k <= ns[e]:
                L = cp.likelihood(cp.BINOMIAL_PRECESSION, g["bin_x"], int(k), t=g["bin_t"][e], n_meas=int(ns[e]))
                np.testing.assert_allclose(L, g["bin_L"][i, :, e], rtol=1e-9, atol=1e-300)


def test_g8_binomial_rb_likelihood(golden):
    g = golden("g8_binomial_rb")
    x, ms, ns = g["brb_x"], g["brb_m"], g["brb_n"]
    for k in range(0, 41, 5):
        for e in range(len(ms)):
            L = cp.likelihood(cp.BINOMIAL_RB, x, k, m=int(ms[e]), n_meas=int(ns[e]))
            np.testing.assert_allclose(L, g["brb_L"][k, :, e], rtol=1e-12, atol=1e-300)


def _resampler_draws(g):
    """The recorded draws the resampler consumed (everything after the prior's), flat, in consumption order."""
    kinds, shapes = g["draw_kinds"], g["draw_shapes"]
    skip = 0
    for i in range(int(g["n_prior_draws"])):
        skip += int(np.prod([v for v in shapes[i] if v >= 0]))
    assert not np.any(kinds[int(g["n_prior_draws"]):] == 2)
    return np.ascontiguousarray(g["draw_data"][skip:])


def _replay_traj(g, kind, cond, horizon_only=True, **ep):
    K = len(g["outcomes"])
    if horizon_only:
        ok = [tol.well_conditioned(float(cond(k))) for k in range(K)]
        K = ok.index(False) if False in ok else K
    assert K >= min(len(g["outcomes"]), 60)
    check_every = ep.pop("check_every", 1)
    ep = {k: v[:K] for k, v in ep.items()}
    r = cp.smc_run(kind, g["x0"], g["outcomes"][:K], rng_mode=2, replay=_resampler_draws(g), threads=1,
                   want_means=True, check_every=check_every, **ep)
    assert r["rc"] == 0
    # resample count per datum is implied by where n_ess jumps back to N; compare the final count and every record
    assert r["resample_count"] == g["resample_count"][K - 1]
    for k in range(K):
        c = float(cond(k))
        np.testing.assert_allclose(r["norms"][k], g["norms"][k], rtol=tol.rtol_norm(c), err_msg="datum %d" % k)
        np.testing.assert_allclose(r["means"][k], g["means"][k], rtol=0, atol=tol.atol_mean(g["means"][k]),
                                   err_msg="datum %d" % k)
    if K == len(g["outcomes"]):
        assert r["replay_used"] == len(_resampler_draws(g))          # every recorded draw consumed, none left over
        np.testing.assert_allclose(r["locs"], g["final_locs"], rtol=1e-9, atol=1e-13)
    return r, K


@pytest.mark.parametrize("name", ["g1_precession_n1000", "g1_precession_n256"])
def test_g1_precession_replay(golden, name):
    g = golden(name)
    _replay_traj(g, cp.PRECESSION, lambda k: g["ep_t"][k], t=g["ep_t"])


def test_g1_precession_batch5_replay(golden):
    g = golden("g1_precession_batch5")
    _replay_traj(g, cp.PRECESSION, lambda k: g["ep_t"][k], t=g["ep_t"], check_every=5)


def test_g1_binomial_replay(golden):
    g = golden("g1_binomial_n1000")
    _replay_traj(g, cp.BINOMIAL_PRECESSION, lambda k: 25 * g["ep_x"][k], t=g["ep_x"], n_meas=g["ep_n_meas"])


def test_g1_rb_replay(golden):
    g = golden("g1_rb_n2000")
    _replay_traj(g, cp.RB, lambda k: g["ep_m"][k], m=g["ep_m"])


def test_g8_binomial_rb_replay(golden):
    g = golden("g8_binomial_rb_n1500")
    _replay_traj(g, cp.BINOMIAL_RB, lambda k: 25 * g["ep_m"][k], m=g["ep_m"], n_meas=g["ep_n_meas"])


def test_g1_tomography_replay(golden):
    g = golden("g1_tomography_n300")
    basis = orc.pauli_data(2)
    K = len(g["outcomes"])
    r = cp.smc_run(cp.TOMOGRAPHY, g["x0"], g["outcomes"], meas=g["ep_meas"], rng_mode=2, replay=_resampler_draws(g),
                   threads=1, basis=basis, want_means=True)
    assert r["rc"] == 0 and r["resample_count"] == g["resample_count"][K - 1]
    for k in range(K):
        at = tol.atol_sqrtm_psd(g["covs"][k])
        np.testing.assert_allclose(r["means"][k], g["means"][k], rtol=0, atol=at)
    np.testing.assert_allclose(r["locs"], g["final_locs"], rtol=0, atol=10 * at)


def test_seeded_stream_matches_numpy_oracle():
    """MT mode consumes np.random's stream in the reference's order: same resamples as the NumPy oracle on one seed."""
    n, K = 2000, 70
    ts = (9 / 8) ** np.arange(K)
    rs = np.random.RandomState(3)
    x0 = rs.random_sample((n, 1))
    outcomes = (rs.random_sample(K) >= np.cos(0.3 * ts / 2) ** 2).astype(int)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.random.seed(11)
        ref = orc.OracleSMC(orc.precession_model(), n, lambda m: x0.copy())
        means = []
        for k in range(K):
            ref.update(int(outcomes[k]), {"t": ts[k:k + 1]})
            means.append(ref.est_mean()[0])
    r = cp.smc_run(cp.PRECESSION, x0, outcomes, t=ts, rng_mode=0, seed=11, threads=1, want_means=True)
    assert r["resample_count"] == ref.resample_count
    for k in range(K):
        np.testing.assert_allclose(r["norms"][k], ref.normalization_record[k], rtol=tol.rtol_norm(ts[k]))
        np.testing.assert_allclose(r["means"][k, 0], means[k], rtol=0, atol=1e-9)


def test_threads_do_not_change_the_algorithm():
    """Philox mode: the draws are keyed by particle, so 1 thread and several give the same cloud up to the rounding
    of the parallel sums."""
    n, K = 20000, 40
    ts = (9 / 8) ** np.arange(K)
    rs = np.random.RandomState(5)
    x0 = rs.random_sample((n, 1))
    outcomes = (rs.random_sample(K) >= np.cos(0.3 * ts / 2) ** 2).astype(int)
    r1 = cp.smc_run(cp.PRECESSION, x0, outcomes, t=ts, rng_mode=1, seed=7, threads=1)
    r4 = cp.smc_run(cp.PRECESSION, x0, outcomes, t=ts, rng_mode=1, seed=7, threads=4)
    assert r1["rc"] == 0 and r4["rc"] == 0 and r4["threads"] == 4
    assert r1["resample_count"] == r4["resample_count"] > 3
    np.testing.assert_allclose(r4["mean"], r1["mean"], rtol=0, atol=1e-9)
    assert abs(r1["mean"][0] - 0.3) < 0.01
