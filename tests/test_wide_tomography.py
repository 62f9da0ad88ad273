"""GPU parity tests of the wide path (round 6): tomography beyond two qubits -- dim 5 .. 8, d = dim^2 up to 64, three
qubits -- through csrc/kernels/wide.hpp.  The reference takes any dim (tomography/models.py:82-226); fixtures
g1_tomography_3q_n200, g2_tomography_wide, g3_moments_wide, g5_canonicalize_wide were generated from it
(oracle/gen_golden.py).  Same stated tolerances as the narrow path (tests/parity_tols.py)."""
import warnings

import numpy as np
import pytest

import np_oracle as orc
import parity_tols as tol
from test_gpu_parity import Replay, fixed_prior, _two_sample_checks, STAT_ALPHA    # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qi():
    import qinfer_amd
    return qinfer_amd


@pytest.fixture(scope="module")
def eng(qi):
    from qinfer_amd.engine import get_engine
    return get_engine()


@pytest.fixture(autouse=True)
def _reset_test_hooks(qi):
    yield
    qi._native.test_hook("canon_wide_jacobi", 0.0)


def _bases(qi):
    t = qi.tomography
    return {"3q": t.pauli_basis(3), "gm5": t.gell_mann_basis(5), "gm7": t.gell_mann_basis(7),
            "q2xq3": t.tensor_product_basis(t.gell_mann_basis(2), t.gell_mann_basis(3))}


def test_wide_models_are_native(qi):
    for tag, b in _bases(qi).items():
        m = qi.TomographyModel(b)
        assert m._native and m._native_canonicalize_ok(), tag
    assert not qi.TomographyModel(qi.tomography.gell_mann_basis(9))._native        # d = 81: beyond the wide kernels


def test_wide_likelihood_g2(qi, golden):
    g = golden("g2_tomography_wide")
    for tag, b in _bases(qi).items():
        if tag + "_x" not in g.files:
            continue
        np.testing.assert_array_equal(b.data, g[tag + "_basis"])
        m = qi.TomographyModel(b)
        ep = np.zeros((g[tag + "_meas"].shape[0],), dtype=m.expparams_dtype)
        ep["meas"] = g[tag + "_meas"]
        L = m.likelihood(np.array([0, 1]), g[tag + "_x"], ep)
        assert L.shape == g[tag + "_L"].shape
        # a sum of up to 64 products: the reference's einsum and the kernel's ascending sum differ by rounding only
        np.testing.assert_allclose(L, g[tag + "_L"], rtol=0, atol=64 * tol.EPS, err_msg=tag)


@pytest.mark.parametrize("n", [1, 7, 2047, 2049, 100003])
def test_wide_update_vs_oracle(qi, n):
    """SMCUpdater.update through k_update_tomo_wide (both tile forms, implicit and explicit weights, sparse and dense
    measurement vectors) against np_oracle: normalisation, n_ess, weights."""
    b = qi.tomography.pauli_basis(3)
    m = qi.TomographyModel(b)
    rs = np.random.RandomState(n)
    x0 = orc.ginibre_prior_sample(n, b.data, rs)
    eps = []
    for k in range(6):
        ep = np.zeros((1,), dtype=m.expparams_dtype)
        if k % 3 == 2:                                   # dense: a random projector's coefficients
            v = rs.randn(8) + 1j * rs.randn(8)
            v /= np.linalg.norm(v)
            ep["meas"][0] = np.real(np.einsum('aij,ij->a', b.data.conj(), np.outer(v, v.conj())))
        else:
            # (I +- P) / 2 = (sqrt 8 / 2) (B_0 +- B_p) in the orthonormal basis B_a = P_a / sqrt 8
            ep["meas"][0, 0] = np.sqrt(8) / 2
            ep["meas"][0, rs.randint(1, 64)] = (1.0 if k % 2 else -1.0) * np.sqrt(8) / 2
        eps.append(ep)
    outs = [int(rs.random_sample() < 0.5) for _ in eps]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        upd = qi.SMCUpdater(m, n, fixed_prior(qi, x0))
        assert upd._native
        w = np.ones(n) / n
        for k, ep in enumerate(eps):
            upd.update(outs[k], ep, check_for_resample=False)
            L = orc.lik_tomography([outs[k]], x0, ep["meas"])[0, :, 0]
            wn = w * L
            norm = wn.sum()
            np.testing.assert_allclose(upd.normalization_record[-1], norm, rtol=1e-12, err_msg="datum %d" % k)
            w = wn / norm
            np.testing.assert_allclose(upd.n_ess, 1.0 / np.sum(w * w), rtol=1e-11)
        np.testing.assert_allclose(upd.particle_weights, w, rtol=1e-11, atol=1e-300)
        np.testing.assert_allclose(upd.est_mean(), orc.particle_mean(w, x0), rtol=0, atol=1e-12)
        cov = orc.particle_cov(w, x0, warn=False)
        np.testing.assert_allclose(upd.est_covariance_mtx(), cov, rtol=0, atol=1e-12)


def test_wide_moments_g3(qi, eng, golden):
    """k_moments_wide<2, 3, 4> (upper block triangle on the f64 matrix cores) against the reference's particle_mean /
    particle_covariance_mtx (utils.py:216-287) at d = 17 ... 64, ragged sizes."""
    g = golden("g3_moments_wide")
    for tag in g["tags"]:
        w, x = g[tag + "_w"], g[tag + "_x"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pd = qi.ParticleDistribution(particle_locations=x, particle_weights=w)
            mean, cov = pd.est_mean(), pd.est_covariance_mtx()
        scale = np.abs(g[tag + "_mean"]).max() ** 2 + np.einsum('i,ij->', w, x * x)
        np.testing.assert_allclose(mean, g[tag + "_mean"], rtol=0, atol=tol.atol_mean(g[tag + "_mean"]), err_msg=tag)
        np.testing.assert_allclose(cov, g[tag + "_cov"], rtol=0, atol=64 * tol.EPS * scale, err_msg=tag)
        S, err = eng.sqrtm_psd(cov)
        np.testing.assert_allclose(S @ S, cov, rtol=0, atol=1e-11 * max(1.0, np.abs(cov).max()), err_msg=tag)


def test_wide_canonicalize_g5(qi, golden):
    g = golden("g5_canonicalize_wide")
    for tag, b in _bases(qi).items():
        x = g[tag + "_x"]
        dim = b.dim
        y = qi.TomographyModel(b).canonicalize(x)
        np.testing.assert_allclose(y, g[tag + "_y"], rtol=0, atol=1e-12, err_msg=tag)
        y2 = qi.TomographyModel(b, allow_subnormalized=True).canonicalize(x)
        np.testing.assert_allclose(y2, g[tag + "_y_subnorm"], rtol=0, atol=1e-12, err_msg=tag)
        rho = np.tensordot(y, b.data, 1)
        np.testing.assert_allclose(np.trace(rho, axis1=1, axis2=2).real, 1.0, atol=1e-12)
        assert np.linalg.eigvalsh(rho).min() > -1e-12
        # a larger cloud: every lane of several waves, both verdicts of the classify pass
        rs = np.random.RandomState(dim)
        xx = orc.ginibre_prior_sample(3001, b.data, rs)
        xx[::2, 1:] += (0.25 / dim) * rs.randn(1501, dim * dim - 1)
        np.testing.assert_allclose(qi.TomographyModel(b).canonicalize(xx), orc.tomo_canonicalize(xx, b.data), rtol=0,
                                   atol=1e-12, err_msg=tag)


def test_wide_canonicalize_two_forms(qi, golden):
    """The listed particles of a dim 5 .. 8 canonicalize through both forms: four lanes per particle, one-sided Jacobi,
    (A + |A|) / 2 (the default) and one lane per particle, eigenvector Jacobi (qsmc_test_hook canon_wide_jacobi) -- each
    against the reference's fixture, and against each other."""
    g = golden("g5_canonicalize_wide")
    for tag, b in _bases(qi).items():
        if b.dim not in (5, 8):                # (the eigenvector form is built for dim 5 and 8)
            continue
        x = g[tag + "_x"]
        rs = np.random.RandomState(b.dim)
        xx = orc.ginibre_prior_sample(2000, b.data, rs)
        xx[:, 1:] += (0.3 / b.dim) * rs.randn(2000, b.dim ** 2 - 1)
        # rank-deficient states: zero eigenvalues and clusters (the degenerate corner of the one-sided form)
        pure = orc.ginibre_prior_sample(64, b.data, rs)
        v = rs.randn(64, b.dim) + 1j * rs.randn(64, b.dim)
        v /= np.linalg.norm(v, axis=1)[:, None]
        pure = np.real(np.einsum('aij,nij->na', b.data.conj(), v[:, :, None] * v[:, None, :].conj()))
        xx = np.concatenate([xx, pure, pure + 1e-9 * rs.randn(*pure.shape)])
        outs = {}
        for form in ("coop", "jacobi"):
            qi._native.test_hook("canon_wide_jacobi", 1.0 if form == "jacobi" else 0.0)
            m = qi.TomographyModel(b)
            np.testing.assert_allclose(m.canonicalize(x), g[tag + "_y"], rtol=0, atol=1e-12, err_msg=tag + form)
            outs[form] = m.canonicalize(xx)
            np.testing.assert_allclose(outs[form], orc.tomo_canonicalize(xx, b.data), rtol=0, atol=1e-12, err_msg=tag + form)
        qi._native.test_hook("canon_wide_jacobi", 0.0)
        np.testing.assert_allclose(outs["coop"], outs["jacobi"], rtol=0, atol=1e-13, err_msg=tag)


def test_traj_tomography_3q(qi, golden):
    """The reference's three-qubit trajectory (fixture g1_tomography_3q_n200: 240 data, 2 Liu-West resamples with
    canonicalize), its RNG draws replayed through the legacy-RNG path: k_update_tomo_wide, k_moments_wide, the host square
    root, k_centres_wide / k_perturb_wide, the wide canonicalize -- datum by datum."""
    g = golden("g1_tomography_3q_n200")
    m = qi.TomographyModel(qi.tomography.pauli_basis(3))
    n, stride = int(g["n_particles"]), int(g["cov_stride"])
    with Replay(g) as rp, warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = qi.LiuWestResampler(kernel=rp.kernel, default_n_particles=n)
        upd = qi.SMCUpdater(m, n, fixed_prior(qi, g["x0"]), resampler=res, canonicalize=True)
        assert upd._native
        for k in range(len(g["outcomes"])):
            ep = np.zeros((1,), dtype=m.expparams_dtype)
            ep["meas"][0] = g["ep_meas"][k]
            upd.update(g["outcomes"][k], ep)
            assert upd.resample_count == g["resample_count"][k], "datum %d" % k
            at = tol.atol_sqrtm_psd(g["covs"][k // stride])
            np.testing.assert_allclose(upd.normalization_record[-1], g["norms"][k], rtol=1e-12 + 10 * at, err_msg="datum %d" % k)
            np.testing.assert_allclose(upd.n_ess, g["n_ess"][k], rtol=1e-11 + 40 * at)
            np.testing.assert_allclose(upd.est_mean(), g["means"][k], rtol=0, atol=max(at, 1e-12), err_msg="datum %d" % k)
    assert upd.resample_count == 2
    np.testing.assert_allclose(upd.particle_locations, g["final_locs"], rtol=0, atol=10 * at)


@pytest.mark.parametrize("case", ["3q-direct", "gm5-direct", "3q-bucketed", "q2xq3-bucketed"])
def test_wide_liu_west_philox_vs_twin(qi, case):
    """The device-RNG resample of a wide cloud on identical Philox numbers (oracle/philox.py): the direct form (small
    clouds: k_anc_direct + k_kick_wide<NB, true>) and the bucketed one (k_bucket_anc16 + k_kick_wide<NB, false>)."""
    import philox as ph
    tag, form = case.split("-")
    b = _bases(qi)[tag]
    d = b.dim ** 2
    rs = np.random.RandomState(d)
    always = lambda z: np.ones(z.shape[0], dtype=bool)       # noqa: E731
    n, n_out = (1500, 4000) if form == "direct" else (30011, 40000)
    x = orc.ginibre_prior_sample(n, b.data, rs)
    w = rs.random_sample(n) ** 2
    if form == "bucketed":
        w[5000:9200] = 0.0
        w[20000:20100] *= 400.0
    model = qi.TomographyModel(b)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pd = qi.ParticleDistribution(particle_locations=x, particle_weights=w)
        res = qi.LiuWestResampler(a=0.9, device_rng=True, seed=99)
        new = res(model, pd, n_particles=n_out)
        wn = np.asarray(pd.particle_weights)
        if form == "direct":
            ref = ph.liu_west_philox(wn, x, always, 0.9, np.sqrt(1 - 0.81), 99, 1, n_out)[0]
        else:
            ref = ph.liu_west_philox_bucketed(wn, x, always, 0.9, np.sqrt(1 - 0.81), 99, 1, n_out,
                                              z_stride=16 * ((d + 15) // 16))[0]
    got = np.asarray(new.particle_locations)
    cov = orc.particle_cov(wn, x, warn=False)
    at = 1e-12 + tol.atol_sqrtm_psd(cov) * 10
    bad = np.abs(got - ref).max(axis=1) > at
    assert bad.sum() <= tol.max_js_flips(n_out), int(bad.sum())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        again = qi.LiuWestResampler(a=0.9, device_rng=True, seed=99)(model, pd, n_particles=n_out).particle_locations
    np.testing.assert_array_equal(got, np.asarray(again))


def test_wide_device_resampler_vs_pinned_oracle_statistics(qi):
    """The wide perf path tied to the reference's restatement: seeds of the device resample of one weighted three-qubit
    Ginibre cloud through an SMCUpdater (canonicalize on) against seeds of np_oracle.liu_west (G4-pinned) followed by
    np_oracle.tomo_canonicalize (G5-pinned; wide fixtures in this round).  Every device output is a physical state."""
    from scipy import stats
    rs = np.random.RandomState(6)
    n, a = 17000, 0.9
    basis = qi.tomography.pauli_basis(3)
    tm = qi.TomographyModel(basis)
    x = orc.ginibre_prior_sample(n, basis.data, rs)
    w = rs.random_sample(n) ** 2
    w /= w.sum()
    always = lambda z: np.ones(z.shape[0], dtype=bool)      # noqa: E731
    seeds = list(range(201, 213))
    dev, ref = [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        upd = qi.SMCUpdater(tm, n, fixed_prior(qi, x), device_rng=True, seed=0)
        upd.particle_weights = w
        for s_ in seeds:
            upd2 = qi.SMCUpdater(tm, n, fixed_prior(qi, x), resampler=qi.LiuWestResampler(a=a, device_rng=True, seed=s_))
            upd2.particle_weights = w
            upd2.resample()
            dev.append(np.asarray(upd2.particle_locations))
            np.random.seed(1000 + s_)
            kicked = orc.liu_west(w, x, always, orc.LegacyRNG(), a=a)[0]
            ref.append(orc.tomo_canonicalize(kicked, basis.data))
    dev, ref = np.stack(dev), np.stack(ref)
    rho_k = np.einsum("na,aij->nij", kicked, basis.data.conj())
    frac_unphysical = np.mean(np.linalg.eigvalsh((rho_k + rho_k.conj().transpose(0, 2, 1)) / 2).min(axis=1) < 0)
    assert 0.05 < frac_unphysical, frac_unphysical
    flat = dev.reshape(-1, 64)
    assert np.abs(flat[:, 0] - 1 / np.sqrt(8)).max() < 1e-12
    rho = np.einsum("na,aij->nij", flat[::7], basis.data.conj())
    ev = np.linalg.eigvalsh((rho + rho.conj().transpose(0, 2, 1)) / 2)
    assert ev.min() > -1e-12 and np.abs(ev.sum(axis=1) - 1.0).max() < 1e-12
    # same law on the 63 free coordinates: KS over the pooled particles, Welch on per-seed means and variances, and the
    # covariances of 60 fixed coordinate pairs (all 1953 of them at this level would fail by chance too often)
    alpha = STAT_ALPHA / 2
    for q in range(1, 64):
        ks = stats.ks_2samp(dev[:, ::3, q].ravel(), ref[:, ::3, q].ravel())
        assert ks.pvalue > alpha, ("KS", q, ks)
        for what, fn in (("mean", lambda c: c[:, :, q].mean(axis=1)), ("var", lambda c: c[:, :, q].var(axis=1))):
            t = stats.ttest_ind(fn(dev), fn(ref), equal_var=False)
            assert t.pvalue > alpha, (what, q, t)
    prs = np.random.RandomState(1)
    for _ in range(60):
        q, r = prs.choice(np.arange(1, 64), 2, replace=False)
        cv = lambda c: np.array([np.cov(c[k, :, q], c[k, :, r])[0, 1] for k in range(c.shape[0])])   # noqa: E731
        t = stats.ttest_ind(cv(dev), cv(ref), equal_var=False)
        assert t.pvalue > alpha, ("cov", q, r, t)


def test_wide_end_to_end_three_qubits(qi):
    """A three-qubit run on the device generator: 400 random Pauli measurements of a random state, N = 60000 -- resamples
    happen, every particle stays a state, the estimate moves towards the truth like the oracle's (smaller) run."""
    b = qi.tomography.pauli_basis(3)
    m = qi.TomographyModel(b)
    rs = np.random.RandomState(31)
    true = orc.ginibre_prior_sample(1, b.data, rs)[0]
    K = 400
    eps, outs = [], []
    for k in range(K):
        ep = np.zeros((1,), dtype=m.expparams_dtype)
        p = rs.randint(1, 64)
        ep["meas"][0, 0] = np.sqrt(8) / 2
        ep["meas"][0, p] = np.sqrt(8) / 2
        eps.append(ep)
        outs.append(int(rs.random_sample() < np.clip(ep["meas"][0] @ true, 0, 1)))
    n = 60000
    np.random.seed(11)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        upd = qi.SMCUpdater(m, n, qi.GinibreDistribution(b), device_rng=True, seed=4)
        prior_err = np.linalg.norm(upd.est_mean() - true)
        for k in range(K):
            upd.update(outs[k], eps[k])
        np.random.seed(12)
        ref = orc.OracleSMC(orc.tomography_model(b.data), 3000, lambda mm: orc.ginibre_prior_sample(mm, b.data, np.random))
        for k in range(K):
            ref.update(outs[k], {"meas": eps[k]['meas']})
    assert upd.resample_count >= 1
    xl = np.asarray(upd.particle_locations)
    rho = np.tensordot(xl[::11], b.data, 1)
    np.testing.assert_allclose(np.trace(rho, axis1=1, axis2=2).real, 1.0, atol=1e-10)
    assert np.linalg.eigvalsh(rho).min() > -1e-10
    err = np.linalg.norm(upd.est_mean() - true)
    assert err < prior_err and err < np.linalg.norm(ref.est_mean() - true) + 0.1
    # batch_update of a wide model: sparse measurement vectors go through the window kernel (k_update_multi_tomo reads rows
    # by index, whatever d is), a window with a dense vector datum by datum -- the per-datum loop's records either way
    from qinfer_amd.engine import get_engine
    eng = get_engine()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a_ = qi.SMCUpdater(m, 5000, qi.GinibreDistribution(b), device_rng=True, seed=5)
        x0 = np.asarray(a_.particle_locations).copy()
        ep10 = np.concatenate(eps[:10])
        dense = ep10.copy()
        v = rs.randn(8) + 1j * rs.randn(8)
        v /= np.linalg.norm(v)
        dense["meas"][7] = np.real(np.einsum('aij,ij->a', b.data.conj(), np.outer(v, v.conj())))
        for eps_w, n_windows in ((ep10, 2), (dense, 1)):
            b1 = qi.SMCUpdater(m, 5000, fixed_prior(qi, x0), device_rng=True, seed=5)
            b2 = qi.SMCUpdater(m, 5000, fixed_prior(qi, x0), device_rng=True, seed=5)
            eng.set_profiling(1)
            b1.batch_update(np.array(outs[:10]), eps_w, resample_interval=5)
            ms, tags = eng.profile_read()
            eng.set_profiling(0)
            assert int(np.sum(tags == 10)) == n_windows, (tags, n_windows)       # QSMC_PROF_UPDATE_MULTI launches
            for k in range(10):
                b2.update(outs[k], eps_w[k:k + 1], check_for_resample=False)
                if k % 5 == 4:
                    b2._maybe_resample()
            np.testing.assert_allclose(b1.normalization_record, b2.normalization_record, rtol=1e-12)
            np.testing.assert_allclose(b1.particle_weights, b2.particle_weights, rtol=1e-11, atol=1e-300)
            assert b1.resample_count == b2.resample_count
    # expected information gain of a wide model goes through the likelihood kernel (no design kernel above d = 16)
    eig = b1.expected_information_gain(ep10[:2])
    assert eig.shape == (2,) and np.all(np.isfinite(eig)) and np.all(eig >= -1e-12)


def test_wide_random_walk_and_kl(qi, eng):
    """Read-outs and decorators on a wide cloud: qsmc_random_walk moves up to 64 rows (host-drawn steps bit for bit, Philox
    steps against the twin), est_kl_divergence takes the reference's own evaluation above d = 16 (against the oracle)."""
    import philox as ph
    rs = np.random.RandomState(9)
    n, d = 3001, 64
    x0 = rs.randn(n, d)
    scale = np.where(np.arange(d) % 3 == 0, 0.0, 0.1 + 0.01 * np.arange(d))
    x = eng.locs_to_soa(x0)
    z = rs.randn(int(np.count_nonzero(scale)), n)
    eng.random_walk(x, scale, z=eng.to_device(z))
    want = x0.copy()
    want[:, scale != 0] += (scale[scale != 0][:, None] * z).T
    np.testing.assert_array_equal(x.cpu().numpy().T, want)
    x = eng.locs_to_soa(x0)
    eng.random_walk(x, scale, seed=77, epoch=3)
    zz = ph.random_walk_normals(n, int(np.count_nonzero(scale)), 77, 3)
    want = x0.copy()
    want[:, scale != 0] += (scale[scale != 0][:, None] * zz).T
    np.testing.assert_allclose(x.cpu().numpy().T, want, rtol=0, atol=1e-13)
    # KL divergence of two wide clouds
    b = qi.tomography.pauli_basis(3)
    xa = orc.ginibre_prior_sample(300, b.data, rs)
    xb = orc.ginibre_prior_sample(200, b.data, rs)
    wa, wb = rs.random_sample(300), rs.random_sample(200)
    wa /= wa.sum()
    wb /= wb.sum()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = qi.ParticleDistribution(particle_locations=xa, particle_weights=wa)
        q = qi.ParticleDistribution(particle_locations=xb, particle_weights=wb)
        got = p.est_kl_divergence(q, delta=0.5)
    np.testing.assert_allclose(got, orc.kl_divergence(xa, wa, xb, wb, delta=0.5), rtol=1e-10)


@pytest.mark.parametrize("counts", [[12000, 20000, 8000], [0, 9000, 9000, 1], [2999]])
def test_wide_sharded_resample_placement(qi, eng, counts):
    """qsmc_lw_resample_philox_sharded of a wide cloud == the single-cloud sampler's particles dealt to the destination
    ranks with exact quotas (AoS rows grouped by destination): k_kick_wide through OutPlace, bucketed and direct forms."""
    from test_gpu_parity import _deal_rows
    b = qi.tomography.pauli_basis(3)
    rs = np.random.RandomState(15)
    n = 30000
    x = orc.ginibre_prior_sample(n, b.data, rs)
    w = rs.random_sample(n) ** 2
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pd = qi.ParticleDistribution(particle_locations=x, particle_weights=w)
        mean, cov = pd.est_mean(), pd.est_covariance_mtx()
    S, _ = eng.sqrtm_psd(cov, scale=0.3)
    desc = qi.TomographyModel(b)._native_desc()
    n_out = int(np.sum(counts))
    ref, f1 = eng.lw_resample_philox(desc, True, pd._x, pd._w, 1.0, 0.95, mean, S, n_out, 77, 3, 1000)
    rows, f2 = eng.lw_resample_philox_sharded(desc, True, pd._x, pd._w, 1.0, 0.95, mean, S, counts, 77, 3, 1000)
    ref = ref.cpu().numpy().T
    rows = rows.cpu().numpy()
    assert rows.shape == (n_out, 64) and f1 == f2 == 0
    place = _deal_rows(counts)
    np.testing.assert_array_equal(rows[place], ref)


def test_ginibre_prior_device_draw(qi):
    """`GinibreDistribution(basis, device=True)` under `SMCUpdater(device_rng=True)`: the prior drawn on the GPU (bench's
    full-size C5: 1e7 states take half a minute on the host).  The Ginibre prior is unpinned (qutip is absent):
    invariants -- every particle a state (trace 1, positive), the same seed the same cloud, another seed another -- and
    the ensemble's moments against the host draw's (the restated Ginibre the oracle tests use) within sampling error."""
    for basis in (qi.tomography.pauli_basis(2), qi.tomography.pauli_basis(3), qi.tomography.gell_mann_basis(3)):
        m = qi.TomographyModel(basis)
        d = basis.dim ** 2
        n = 40000
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = qi.SMCUpdater(m, n, qi.GinibreDistribution(basis, device=True), device_rng=True, seed=7)
            b = qi.SMCUpdater(m, n, qi.GinibreDistribution(basis, device=True), device_rng=True, seed=7)
            c = qi.SMCUpdater(m, n, qi.GinibreDistribution(basis, device=True), device_rng=True, seed=8)
        xa, xb, xc = (np.asarray(u.particle_locations) for u in (a, b, c))
        assert xa.shape == (n, d)
        assert np.array_equal(xa, xb) and not np.array_equal(xa, xc)
        rho = np.tensordot(xa[::7], basis.data, 1)
        np.testing.assert_allclose(np.trace(rho, axis1=1, axis2=2).real, 1.0, atol=1e-12)
        assert np.linalg.eigvalsh(rho).min() > -1e-12
        np.random.seed(3)
        xh = qi.GinibreDistribution(basis).sample(n)
        se = np.sqrt((xa.var(0) + xh.var(0)) / n) + 1e-15
        assert np.all(np.abs(xa.mean(0) - xh.mean(0)) < 5 * se)
        v_a, v_h = xa[:, 1:].var(0), xh[:, 1:].var(0)
        assert np.all(np.abs(v_a / v_h - 1) < 0.08), (v_a / v_h)


@pytest.mark.parametrize("make_basis", [lambda q: q.tomography.pauli_basis(1), lambda q: q.tomography.gell_mann_basis(3),
                                        lambda q: q.tomography.pauli_basis(2), lambda q: q.tomography.gell_mann_basis(6),
                                        lambda q: q.tomography.pauli_basis(3)],
                         ids=["pauli-1q", "gell-mann-3", "pauli-2q", "gell-mann-6", "pauli-3q"])
def test_tomography_design_quantities_every_dim(qi, make_basis):
    """`hypothetical_update`, `bayes_risk` and `expected_information_gain` of a tomography model against their definitions
    (smc.py:324-386, 553-663) evaluated in NumPy on the cloud's host snapshot, dense and sparse measurement vectors, for
    dim 2 ... 8.  (A ONE-qubit model, d = 4, used to read its design rows with the moment layout of the d <= 4 kernels --
    tomography's kernels carry no moments: rows of two columns; bayes_risk came back negative / -inf.)"""
    basis = make_basis(qi)
    m = qi.TomographyModel(basis)
    rs = np.random.RandomState(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.random.seed(1)
        u = qi.SMCUpdater(m, 3000, qi.GinibreDistribution(basis), device_rng=True, seed=2)
        eps = np.zeros(5, dtype=m.expparams_dtype)
        for i in range(3):
            v = rs.randn(basis.dim) + 1j * rs.randn(basis.dim)
            v /= np.linalg.norm(v)
            eps['meas'][i] = np.real(np.einsum('aij,ij->a', basis.data.conj(), np.outer(v, v.conj())))
        for i in range(3, 5):
            eps['meas'][i, 0] = eps['meas'][i, i - 2] = np.sqrt(basis.dim) / 2
        for k in range(5):
            u.update(int(rs.randint(2)), eps[k:k + 1])
        x, w = np.asarray(u.particle_locations), np.asarray(u.particle_weights)
        pr1 = np.clip(x @ eps['meas'].T, 0, 1)
        L = np.stack([1 - pr1, pr1]).transpose(0, 2, 1)                       # (2, n_e, N)
        norm_ref = (L * w).sum(-1, keepdims=True)
        hw, norm = u.hypothetical_update(np.array([0, 1]), eps, return_normalization=True)
        np.testing.assert_allclose(norm, norm_ref, rtol=1e-10)
        np.testing.assert_allclose(hw, L * w / norm_ref, rtol=1e-8, atol=1e-300)
        risk_ref, eig_ref = [], []
        for e in range(5):
            r = g = 0.0
            for o in range(2):
                hwe = L[o, e] * w / norm_ref[o, e, 0]
                mu = hwe @ x
                r += norm_ref[o, e, 0] * float((m.Q * (hwe @ (x - mu) ** 2)).sum())
                nz = hwe > 0
                g += norm_ref[o, e, 0] * float((hwe[nz] * np.log(hwe[nz] / w[nz])).sum())
            risk_ref.append(r)
            eig_ref.append(g)
        np.testing.assert_allclose(np.ravel(u.bayes_risk(eps)), risk_ref, rtol=1e-7)
        np.testing.assert_allclose(np.ravel(u.expected_information_gain(eps)), eig_ref, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("which", ["gell-mann-9", "pauli-4q"])
def test_tomography_beyond_the_kernels_through_device_hooks(qi, which):
    """d > 64 (dim 9: d = 81; FOUR qubits: d = 256) has no kernels of the library's own: the model then offers the plugin
    surface's device hooks (`likelihood_device`, `canonicalize_device`: torch on the (d, N) cloud) and the engine forms
    moments, centres and kicks of that size by torch products -- the cloud never leaves the GPU.  Against the restatement
    of the reference on the same draws: 30 updates (norms, mean to rounding), a resample with canonicalize (locations to
    1e-8: two eigensolvers), physical states, further updates, a window, the design quantities."""
    basis = qi.tomography.gell_mann_basis(9) if which == "gell-mann-9" else qi.tomography.pauli_basis(4)
    m = qi.TomographyModel(basis)
    d, n = basis.dim ** 2, 250
    assert not m._native and m.likelihood_device is not None and m.canonicalize_device is not None
    assert getattr(qi.TomographyModel(qi.tomography.pauli_basis(3)), "likelihood_device", None) is None     # (native: its kernels)
    rs = np.random.RandomState(3)
    np.random.seed(1)
    true = qi.GinibreDistribution(basis).sample(1)[0]
    eps, outs = [], []
    for k in range(30):
        ep = np.zeros(1, dtype=m.expparams_dtype)
        ep['meas'][0, 0] = ep['meas'][0, 1 + rs.randint(d - 1)] = np.sqrt(basis.dim) / 2
        eps.append(ep)
        outs.append(int(rs.random_sample() < np.clip(ep['meas'][0] @ true, 0, 1)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.random.seed(2)
        u = qi.SMCUpdater(m, n, qi.GinibreDistribution(basis))
        x0 = np.asarray(u.particle_locations).copy()
        o = orc.OracleSMC(orc.tomography_model(basis.data), n, lambda mm: x0.copy())
        for k in range(30):
            u.update(outs[k], eps[k], check_for_resample=False)
            o.update(outs[k], {"meas": eps[k]['meas']}, check_for_resample=False)
        np.testing.assert_allclose(np.ravel(u.normalization_record), np.ravel(o.normalization_record), rtol=1e-12)
        np.testing.assert_allclose(u.est_mean(), o.est_mean(), rtol=0, atol=1e-13)
        assert u.resample_count == o.resample_count
        np.random.seed(11)
        u.resample()
        np.random.seed(11)
        o.resample()
        xn = np.asarray(u.particle_locations)
        np.testing.assert_allclose(xn, o.x, rtol=0, atol=1e-8)
        rho = np.tensordot(xn, basis.data, 1)
        np.testing.assert_allclose(np.trace(rho, axis1=1, axis2=2).real, 1.0, atol=1e-12)
        assert np.linalg.eigvalsh(rho).min() > -1e-12
        for k in range(5):
            u.update(outs[k], eps[k])
            o.update(outs[k], {"meas": eps[k]['meas']})
        np.testing.assert_allclose(u.est_mean(), o.est_mean(), rtol=0, atol=1e-9)
        u.batch_update(np.array(outs[:6]), np.concatenate(eps[:6]), resample_interval=3)
        assert np.all(np.isfinite(u.bayes_risk(np.concatenate(eps[:2])))) and np.all(
            np.asarray(u.expected_information_gain(np.concatenate(eps[:2]))) >= -1e-12)
