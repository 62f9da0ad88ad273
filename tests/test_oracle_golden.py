"""Pins the CPU oracle (oracle/np_oracle.py) against golden vectors produced by running the
reference itself (oracle/gen_golden.py): likelihood KATs (G2), moments (G3), Liu-West incl.
quirk Q1 (G4), tomography canonicalize (G5), guards (G6) and full seeded trajectories (G1)."""
import warnings

import numpy as np
import pytest

import np_oracle as orc
import parity_tols as tol


def _replay(g, prefix=""):
    return orc.ReplayRNG(g[prefix + "draw_kinds"], g[prefix + "draw_shapes"], g[prefix + "draw_data"])


# ------------------------------------------------------------------ G2
def test_g2_precession(golden):
    g = golden("g2_likelihoods")
    L = orc.lik_precession([0, 1], g["prec_x"], g["prec_t"])
    np.testing.assert_array_equal(L, g["prec_L"])      # same NumPy cos on the same machine class


def test_g2_binomial(golden):
    g = golden("g2_likelihoods")
    L = orc.lik_binomial_precession(np.arange(26), g["bin_x"], g["bin_t"], g["bin_n"])
    ref = g["bin_L"]
    # closed form vs SciPy/Boost: rtol 1e-12 with an absolute floor for denormal-range values
    np.testing.assert_allclose(L, ref, rtol=1e-12, atol=1e-300)


def test_g2_rb(golden):
    g = golden("g2_likelihoods")
    np.testing.assert_array_equal(orc.lik_rb([0, 1], g["rb_x"], g["rb_m"]), g["rb_L"])
    np.testing.assert_array_equal(orc.valid_rb(g["rb_valid_x"]), g["rb_valid"])
    np.testing.assert_array_equal(orc.lik_rb([0, 1], g["rbi_x"], g["rbi_m"], g["rbi_ref"]), g["rbi_L"])
    np.testing.assert_array_equal(orc.valid_rb(g["rbi_x"]), g["rbi_valid"])


def test_g2_tomography_and_bases(golden):
    g = golden("g2_likelihoods")
    np.testing.assert_allclose(orc.pauli_data(2), g["pauli2_basis"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(orc.gell_mann_data(3), g["gell_mann3_basis"], rtol=0, atol=1e-15)
    L = orc.lik_tomography([0, 1], g["tomo_x"], g["tomo_meas"])
    np.testing.assert_allclose(L, g["tomo_L"], rtol=0, atol=4e-16)


def test_g2_edges(golden):
    """Corners: cos arguments beyond 1e10 rad, non-finite parameters, binomial pmfs with many measurements."""
    g = golden("g2_edges")
    with np.errstate(invalid="ignore"):
        L = orc.lik_precession([0, 1], g["prec_x"], g["prec_t"])
    np.testing.assert_array_equal(L, g["prec_L"])              # (NaN == NaN under assert_array_equal)
    ks, ns = g["bin_k"], g["bin_n"]
    for e in range(len(ns)):
        ok = ks <= ns[e]
        L = orc.lik_binomial_precession(ks[ok], g["bin_x"], g["bin_t"][e:e + 1], ns[e:e + 1])[:, :, 0]
        np.testing.assert_allclose(L, g["bin_L"][ok, :, e], rtol=1e-9, atol=1e-300, err_msg="n_meas %d" % ns[e])


# ------------------------------------------------------------------ G3
def test_g3_moments(golden):
    g = golden("g3_moments")
    for tag in g["tags"]:
        w, x = g[tag + "_w"], g[tag + "_x"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mean, cov = orc.particle_mean(w, x), orc.particle_cov(w, x)
        scale = np.abs(mean).max() ** 2 + np.einsum('i,ij->', w, x * x)
        np.testing.assert_allclose(mean, g[tag + "_mean"], rtol=1e-14, atol=1e-16)
        np.testing.assert_allclose(cov, g[tag + "_cov"], rtol=0, atol=8 * orc.EPS * scale)
        np.testing.assert_allclose(orc.n_ess(w), g[tag + "_ess"], rtol=1e-14)
        S, err = orc.sqrtm_psd(cov)
        np.testing.assert_allclose(S, g[tag + "_sqrt"], rtol=1e-12, atol=1e-15)
        assert abs(err - g[tag + "_sqrt_err"]) <= 1e-12 * max(1.0, abs(err))


# ------------------------------------------------------------------ G4
def _model_for(tag, g):
    if tag.startswith("rb"):
        return orc.valid_rb
    if tag.startswith("tomo"):
        return lambda x: np.ones(x.shape[0], dtype=bool)
    return orc.valid_precession


def test_g4_liu_west(golden):
    g = golden("g4_liu_west")
    for tag in g["tags"]:
        rng = _replay(g, tag + "_")
        h = g[tag + "_h"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            new, w = orc.liu_west(g[tag + "_w"], g[tag + "_x"], _model_for(tag, g), rng,
                                  a=float(g[tag + "_a"]), h=None if np.isnan(h) else float(h),
                                  n_out=int(g[tag + "_n_out"]))
        assert rng.exhausted, tag
        np.testing.assert_allclose(new, g[tag + "_new"], rtol=1e-13, atol=1e-15, err_msg=tag)


def test_g4_q1_quirk_is_exercised(golden):
    """The fixed (non-legacy) redraw centres give a different answer on the Q1 fixture."""
    g = golden("g4_liu_west")
    tag = "q1_small"
    rng = _replay(g, tag + "_")
    n_rounds = int(np.sum(g[tag + "_draw_kinds"] == 1))
    assert n_rounds >= 2, "fixture must contain at least one redraw round"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        new, _ = orc.liu_west(g[tag + "_w"], g[tag + "_x"], orc.valid_precession, rng,
                              a=float(g[tag + "_a"]), legacy_mus_truncation=False)
    assert not np.allclose(new, g[tag + "_new"])


# ------------------------------------------------------------------ G5
def test_g5_canonicalize(golden):
    g = golden("g5_canonicalize")
    y = orc.tomo_canonicalize(g["x"], g["basis"])
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=1e-13)
    y2 = orc.tomo_canonicalize(g["x"], g["basis"], allow_subnormalized=True)
    np.testing.assert_allclose(y2, g["y_subnorm"], rtol=0, atol=1e-13)


def test_g5_canonicalize_qutrit(golden):
    """The same restatement on a qutrit (gell_mann_basis(3), d = 9), against the reference's outputs."""
    g = golden("g5_canonicalize_qutrit")
    assert g["x"].shape == (256, 9) and g["basis"].shape == (9, 3, 3)
    np.testing.assert_allclose(orc.tomo_canonicalize(g["x"], g["basis"]), g["y"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(orc.tomo_canonicalize(g["x"], g["basis"], allow_subnormalized=True), g["y_subnorm"],
                               rtol=0, atol=1e-13)


# ------------------------------------------------------------------ G1 trajectories
def _traj(g, model, ep_of, cond, batch=None):
    rng = _replay(g)
    n = int(g["n_particles"])
    x0 = g["x0"]
    def prior(nn):
        # the reference's prior draws are the first n_prior_draws entries of the log
        for _ in range(int(g["n_prior_draws"])):
            (rng.random if rng.kinds[rng.pos] == 0 else rng.randn)(*([rng.shapes[rng.pos]] if rng.kinds[rng.pos] == 0 else rng.shapes[rng.pos]))
        return x0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        smc = orc.OracleSMC(model, n, prior, rng=rng, canonicalize=False)
        smc.x[:, :] = x0
        K = len(g["outcomes"])
        checked = -1
        for k in range(K):
            if not tol.well_conditioned(float(cond(k))):
                break               # see parity_tols.HORIZON_RTOL: compare only while well conditioned
            o = g["outcomes"][k]
            if batch is None:
                smc.update(o, ep_of(k))
            else:
                smc.update(o, ep_of(k), check_for_resample=False)
                if (k + 1) % batch == 0:
                    smc._maybe_resample()
            c = float(cond(k))
            checked = k
            assert smc.resample_count == g["resample_count"][k], "datum %d" % k
            np.testing.assert_allclose(smc.normalization_record[-1], g["norms"][k],
                                       rtol=tol.rtol_norm(c), err_msg="datum %d" % k)
            np.testing.assert_allclose(smc.n_ess, g["n_ess"][k], rtol=tol.rtol_ess(c))
            np.testing.assert_allclose(smc.est_mean(), g["means"][k], rtol=0,
                                       atol=tol.atol_mean(g["means"][k]))
    assert checked >= min(K - 1, 60), "horizon too short to be a meaningful check"
    if checked == K - 1:
        assert rng.exhausted
        np.testing.assert_allclose(smc.x, g["final_locs"], rtol=1e-9, atol=1e-13)
    return smc


@pytest.mark.parametrize("name", ["g1_precession_n1000", "g1_precession_n256"])
def test_g1_precession(golden, name):
    g = golden(name)
    _traj(g, orc.precession_model(), lambda k: {"t": g["ep_t"][k:k + 1]}, lambda k: g["ep_t"][k])


def test_g1_precession_batch5(golden):
    g = golden("g1_precession_batch5")
    _traj(g, orc.precession_model(), lambda k: {"t": g["ep_t"][k:k + 1]}, lambda k: g["ep_t"][k],
          batch=5)


def test_g1_binomial(golden):
    g = golden("g1_binomial_n1000")
    _traj(g, orc.binomial_precession_model(),
          lambda k: {"t": g["ep_x"][k:k + 1], "n_meas": g["ep_n_meas"][k:k + 1]},
          lambda k: 25 * g["ep_x"][k])


def test_g1_rb(golden):
    g = golden("g1_rb_n2000")
    _traj(g, orc.rb_model(), lambda k: {"m": g["ep_m"][k:k + 1]}, lambda k: g["ep_m"][k])


def test_g8_binomial_rb_likelihood(golden):
    g = golden("g8_binomial_rb")
    L = orc.lik_binomial_rb(np.arange(41), g["brb_x"], g["brb_m"], g["brb_n"])
    np.testing.assert_allclose(L, g["brb_L"], rtol=1e-12, atol=1e-300)
    Li = orc.lik_binomial_rb(np.arange(26), g["brbi_x"], g["brbi_m"], g["brbi_n"], g["brbi_ref"])
    np.testing.assert_allclose(Li, g["brbi_L"], rtol=1e-12, atol=1e-300)
    # every column is a pmf over k = 0..n_meas
    np.testing.assert_allclose(L.sum(axis=0), 1.0, rtol=1e-12)


def test_g8_binomial_rb_traj(golden):
    g = golden("g8_binomial_rb_n1500")
    _traj(g, orc.binomial_rb_model(),
          lambda k: {"m": g["ep_m"][k:k + 1], "n_meas": g["ep_n_meas"][k:k + 1]},
          lambda k: 25 * g["ep_m"][k])


def test_g9_unknown_t2_and_mle_likelihoods(golden):
    g = golden("g9_t2_mle")
    np.testing.assert_array_equal(orc.lik_unknown_t2([0, 1], g["t2_x"], g["t2_t"]), g["t2_L"])
    np.testing.assert_array_equal(orc.valid_unknown_t2(g["t2_valid_x"]), g["t2_valid"])
    mle = orc.mle_model(orc.precession_model(), float(g["mle_prec_gamma"]))
    np.testing.assert_array_equal(mle.lik([0, 1], g["mle_prec_x"], {"t": g["mle_prec_t"]}), g["mle_prec_L"])
    mle = orc.mle_model(orc.binomial_precession_model(), float(g["mle_bin_gamma"]))
    L = mle.lik(np.arange(26), g["mle_bin_x"], {"t": g["mle_bin_t"], "n_meas": g["mle_bin_n"]})
    np.testing.assert_allclose(L, g["mle_bin_L"], rtol=1e-12, atol=1e-300)


def test_g9_unknown_t2_traj(golden):
    g = golden("g9_unknown_t2_n2000")
    _traj(g, orc.unknown_t2_model(), lambda k: {"t": g["ep_t"][k:k + 1]}, lambda k: g["ep_t"][k])


def test_g9_mle_traj(golden):
    g = golden("g9_mle_precession_n1000")
    _traj(g, orc.mle_model(orc.precession_model(), 3.0), lambda k: {"t": g["ep_t"][k:k + 1]},
          lambda k: 3.0 * g["ep_t"][k])


def test_g10_random_walk_traj(golden):
    """Time-step updates (smc.py:447-449) through GaussianRandomWalkModel with fixed diagonal covariance."""
    g = golden("g10_grw_precession_n800")
    model = orc.gaussian_random_walk_model(orc.precession_model(), np.sqrt(2.5e-7))
    smc = _traj(g, model, lambda k: {"t": g["ep_t"][k:k + 1]}, lambda k: g["ep_t"][k])
    g = golden("g10_grw_t2_n800")
    model = orc.gaussian_random_walk_model(orc.unknown_t2_model(), np.sqrt(1e-6), idxs=[0],
                                           scale_mult=lambda e: np.sqrt(e["t"]))
    _traj(g, model, lambda k: {"t": g["ep_t"][k:k + 1]}, lambda k: g["ep_t"][k])


def test_g11_readouts(golden):
    g = golden("g11_readouts")
    w, x = g["w"], g["x"]
    np.testing.assert_allclose(orc.est_entropy(w), g["entropy"], rtol=1e-14)
    for lvl in (50, 95):
        inside, outside = orc.est_credible_region(w, x, level=lvl / 100, return_outside=True)
        np.testing.assert_array_equal(inside, g["cred_%d_inside" % lvl])
        assert outside.shape[0] == int(g["cred_%d_n_outside" % lvl])
    np.testing.assert_array_equal(orc.est_credible_region(w, x, 0.95, modelparam_slice=slice(0, 1)), g["cred_95_slice0"])
    np.testing.assert_array_equal(orc.sample_cloud(w, x, g["sample_u"]), g["sample"])
    ps, pr = orc.posterior_marginal(w, x, 0, res=60)
    np.testing.assert_allclose(ps, g["marg0_ps"], rtol=1e-15)
    np.testing.assert_allclose(pr, g["marg0_pr"], rtol=1e-12, atol=1e-12)
    ps, pr = orc.posterior_marginal(w, x, 1, res=40, smoothing=0.004, range_min=0.0, range_max=0.1)
    np.testing.assert_allclose(pr, g["marg1_pr"], rtol=1e-12, atol=1e-12)


def test_g1_tomography(golden):
    g = golden("g1_tomography_n300")
    basis = orc.pauli_data(2)
    model = orc.tomography_model(basis)
    rng = _replay(g)
    n = int(g["n_particles"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        smc = orc.OracleSMC(model, n, lambda nn: g["x0"], rng=rng, canonicalize=True)
        np.testing.assert_allclose(smc.x, g["x0"], atol=1e-13)
        for k in range(len(g["outcomes"])):
            smc.update(g["outcomes"][k], {"meas": g["ep_meas"][k:k + 1]})
            assert smc.resample_count == g["resample_count"][k]
            at = tol.atol_sqrtm_psd(g["covs"][k])
            np.testing.assert_allclose(smc.est_mean(), g["means"][k], rtol=0, atol=at)
    np.testing.assert_allclose(smc.x, g["final_locs"], rtol=0, atol=10 * at)


# ------------------------------------------------------------------ G6
def test_g6_min_ness(golden):
    g = golden("g6_guards")
    N = int(g["N"])
    dec = orc.OracleModel("decimation", 1, None, lambda x: np.ones(x.shape[0], dtype=bool))

    def lik(o, x, e):
        pr0 = np.ones((x.shape[0], 1)) / 2
        pr0[int(np.ceil(e["alpha"][0] * x.shape[0])):, :] = 0
        return orc._two_outcome(o, pr0)
    dec.lik = lik
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        smc = orc.OracleSMC(dec, N, lambda n: np.random.random((n, 1)), resample_thresh=0.0)
        for k in range(6):
            smc.update(0, {"alpha": np.array([4.0 ** -(k + 1)])})
            assert smc.min_n_ess == g["min_n_ess"][k]
            assert smc.n_ess == g["n_ess"][k]


def test_guards_zero_weight_policies():
    """smc.py:423-436 policy table on an impossible datum."""
    impossible = orc.OracleModel("imp", 1, lambda o, x, e: np.zeros((1, x.shape[0], 1)),
                                 lambda x: np.ones(x.shape[0], dtype=bool))
    mk = lambda pol: orc.OracleSMC(impossible, 16, lambda n: np.zeros((n, 1)), zero_weight_policy=pol)
    with pytest.raises(RuntimeError):
        mk("error").update(0, {})
    s = mk("skip")
    s.update(0, {})
    assert np.all(s.w == 1 / 16) and s.normalization_record == []
    with pytest.warns(orc.ApproximationWarning):
        mk("warn").update(0, {})
    with pytest.raises(ValueError):
        mk("bogus").update(0, {})


def test_c1_statistical_full_run():
    """Config C1 end to end on the oracle with the legacy global RNG (tests/test_precession_model.py
    style acceptance: mean to 2 decimals, small covariance)."""
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        smc = orc.OracleSMC(orc.precession_model(), 1000, lambda n: np.random.random((n, 1)))
        for k in range(200):
            t = np.array([(9 / 8) ** k])
            o = int(np.random.random() >= np.cos(0.3 * t[0] / 2) ** 2)
            smc.update(o, {"t": t})
    assert abs(smc.est_mean()[0] - 0.3) < 1e-5
    assert smc.est_covariance_mtx()[0, 0] < 1e-9
    assert 25 <= smc.resample_count <= 60


# ------------------------------------------------------------------ G7 experiment design
def test_g7_design_oracle(golden):
    g = golden("g7_design")
    lik_p = lambda o, x, e: orc.lik_precession(o, x, e["t"])
    eps = [{"t": g["prec_t"][k:k + 1]} for k in range(len(g["prec_t"]))]
    risk = orc.bayes_risk(g["prec_w"], g["prec_x"], lik_p, np.array([0, 1]), eps)
    eig = orc.expected_information_gain(g["prec_w"], g["prec_x"], lik_p, np.array([0, 1]), eps)
    np.testing.assert_allclose(risk, g["prec_risk"], rtol=1e-12)
    np.testing.assert_allclose(eig, g["prec_eig"], rtol=1e-11, atol=1e-15)
    lik_rb = lambda o, x, e: orc.lik_rb(o, x, e["m"])
    eps = [{"m": g["rb_m"][k:k + 1]} for k in range(3)]
    np.testing.assert_allclose(orc.bayes_risk(g["rb_w"], g["rb_x"], lik_rb, np.array([0, 1]), eps),
                               g["rb_risk"], rtol=1e-12)
    np.testing.assert_allclose(orc.expected_information_gain(g["rb_w"], g["rb_x"], lik_rb, np.array([0, 1]), eps),
                               g["rb_eig"], rtol=1e-10, atol=1e-15)
    lik_b = lambda o, x, e: orc.lik_binomial_precession(o, x, e["t"], e["n_meas"])
    for k in range(3):
        e = [{"t": g["bin_t"][k:k + 1], "n_meas": g["bin_n"][k:k + 1]}]
        os_ = np.arange(int(g["bin_n"][k]) + 1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = orc.bayes_risk(g["bin_w"], g["bin_x"], lik_b, os_, e)
        np.testing.assert_allclose(r[0], g["bin_risk"][k], rtol=1e-9)


# ---------------------------------------------------------------------------------------------
# The device-RNG resampler's chunk counts (oracle/philox.py twin of k_bucket_counts): the samplers' laws, checked here against scipy's exact pmfs.
def _chi2_p(xs, cdf, n_bins_edges):
    from scipy import stats
    obs = np.histogram(xs, bins=n_bins_edges + 0.5)[0]
    exp = np.diff(cdf(n_bins_edges)) * len(xs)
    keep = exp > 5
    chi = ((obs - exp) ** 2 / exp)[keep].sum()
    return stats.chi2.sf(chi, keep.sum() - 1)


@pytest.mark.parametrize("mu", [0.7, 6.5, 10.5, 77.0, 4096.0, 3.0e6])
def test_poisson_draw_law(mu):
    """poisson_draw (cdf search below 10, PTRS above) follows Poisson(mu): chi-square over ~30 quantile bins,
    mean and variance, 20000 independent Philox streams."""
    from scipy import stats
    import philox as ph
    M = 20000
    xs = np.array([ph.poisson_draw(mu, node, 991 + int(mu), 2) for node in range(M)])
    qs = np.unique(stats.poisson.ppf(np.linspace(0, 1, 31)[1:-1], mu))
    edges = np.concatenate([[-1], qs, [1e15]])
    assert _chi2_p(xs, lambda e: stats.poisson.cdf(e, mu), edges) > 1e-4
    assert abs(xs.mean() - mu) < 5 * np.sqrt(mu / M)
    assert abs(xs.var() / mu - 1) < 5 * np.sqrt(2.0 / M + 1.0 / (mu * M))


@pytest.mark.parametrize("margin", [5.0, 0.0, -3.0])
def test_poissonised_counts_law(margin):
    """Poisson counts + top-up (margin 5: the production setting) or + removal (margin <= 0 overshoots about half
    the time / nearly always) are Multinomial(n_out; p): exact total, every chunk's count Binomial(n_out, p_c)
    across independent seeds, and the expected covariance sign between two chunks."""
    from scipy import stats
    import philox as ph
    p = np.array([0.02, 0.3, 0.0, 0.08, 0.25, 0.001, 0.349])
    edges = np.cumsum(p)
    edges[-1] = 1.0
    n_out, reps = 3000, 1500
    C = np.array([ph.poissonised_counts(edges, n_out, 1000 + r, 1 + r % 7, margin=margin) for r in range(reps)])
    assert np.all(C.sum(axis=1) == n_out) and C.min() >= 0
    assert np.all(C[:, 2] == 0)
    for c in (0, 1, 3, 4, 5, 6):
        qs = np.unique(stats.binom.ppf(np.linspace(0, 1, 13)[1:-1], n_out, p[c]))
        e = np.concatenate([[-1], qs, [n_out]])
        assert _chi2_p(C[:, c], lambda t: stats.binom.cdf(t, n_out, p[c]), e) > 1e-4, c
    cov = np.cov(C[:, 1], C[:, 4])[0, 1]
    want = -n_out * p[1] * p[4]
    assert abs(cov - want) < 6 * n_out * np.sqrt(p[1] * p[4]) / np.sqrt(reps) + 0.1 * abs(want)
    if margin <= 0:
        lam = n_out - margin * np.sqrt(n_out)
        assert stats.poisson.sf(n_out, lam) > 0.4              # the removal branch really was the common one


def test_g14_kl_divergence(golden):
    """oracle.kl_divergence against the reference's est_kl_divergence and the divergences its updater recorded."""
    g = golden("g14_kl_divergence")
    for tag in ("d1", "d3"):
        x, w, y, v = g[tag + "_x"], g[tag + "_w"], g[tag + "_y"], g[tag + "_v"]
        np.testing.assert_allclose(orc.kl_divergence(x, w, y, v), g[tag + "_kl"], rtol=1e-12)
        np.testing.assert_allclose(orc.kl_divergence(x, w, y, v, delta=0.05), g[tag + "_kl_delta"], rtol=1e-12)
    for tag in ("prec", "rb"):
        assert int(g[tag + "_n_recorded"]) >= 1
        for i in range(int(g[tag + "_n_recorded"])):
            val = orc.kl_divergence(g["%s_r%d_new_x" % (tag, i)], g["%s_r%d_new_w" % (tag, i)],
                                    g["%s_r%d_old_x" % (tag, i)], g["%s_r%d_old_w" % (tag, i)], Q=g[tag + "_Q"])
            np.testing.assert_allclose(val, g["%s_r%d_kl" % (tag, i)], rtol=1e-12)
            np.testing.assert_allclose(val, g[tag + "_divergences"][i], rtol=1e-12)


# ------------------------------------------------------------------ round 6: tomography beyond two qubits (d up to 64)
def test_g2_tomography_wide(golden):
    g = golden("g2_tomography_wide")
    np.testing.assert_allclose(g["3q_basis"], orc.pauli_data(3), atol=1e-15)
    np.testing.assert_allclose(g["gm5_basis"], orc.gell_mann_data(5), atol=1e-15)
    for tag in ("3q", "gm5", "q2xq3"):
        L = orc.lik_tomography([0, 1], g[tag + "_x"], g[tag + "_meas"])
        np.testing.assert_allclose(L, g[tag + "_L"], rtol=0, atol=2e-15, err_msg=tag)


def test_g3_moments_wide(golden):
    g = golden("g3_moments_wide")
    for tag in g["tags"]:
        w, x = g[tag + "_w"], g[tag + "_x"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mean, cov = orc.particle_mean(w, x), orc.particle_cov(w, x)
        scale = np.abs(mean).max() ** 2 + np.einsum('i,ij->', w, x * x)
        np.testing.assert_allclose(mean, g[tag + "_mean"], rtol=1e-14, atol=1e-16)
        np.testing.assert_allclose(cov, g[tag + "_cov"], rtol=0, atol=8 * orc.EPS * scale)
        S, err = orc.sqrtm_psd(cov)
        np.testing.assert_allclose(S, g[tag + "_sqrt"], rtol=0, atol=1e-12 * max(1.0, np.abs(S).max()))


def test_g5_canonicalize_wide(golden):
    g = golden("g5_canonicalize_wide")
    for tag in ("gm5", "q2xq3", "gm7", "3q"):
        x, basis = g[tag + "_x"], g[tag + "_basis"]
        np.testing.assert_allclose(orc.tomo_canonicalize(x, basis), g[tag + "_y"], rtol=0, atol=1e-13, err_msg=tag)
        np.testing.assert_allclose(orc.tomo_canonicalize(x, basis, allow_subnormalized=True), g[tag + "_y_subnorm"], rtol=0,
                                   atol=1e-13, err_msg=tag)


def test_g1_tomography_3q(golden):
    """A three-qubit SMCUpdater trajectory of the reference (d = 64, 240 data, 2 resamples + canonicalize), draws replayed."""
    g = golden("g1_tomography_3q_n200")
    model = orc.tomography_model(orc.pauli_data(3))
    rng = _replay(g)
    n, stride = int(g["n_particles"]), int(g["cov_stride"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        smc = orc.OracleSMC(model, n, lambda nn: g["x0"], rng=rng, canonicalize=True)
        for k in range(len(g["outcomes"])):
            smc.update(g["outcomes"][k], {"meas": g["ep_meas"][k:k + 1]})
            assert smc.resample_count == g["resample_count"][k], "datum %d" % k
            at = tol.atol_sqrtm_psd(g["covs"][k // stride])
            np.testing.assert_allclose(smc.est_mean(), g["means"][k], rtol=0, atol=at, err_msg="datum %d" % k)
    assert smc.resample_count == 2
    np.testing.assert_allclose(smc.x, g["final_locs"], rtol=0, atol=10 * at)
