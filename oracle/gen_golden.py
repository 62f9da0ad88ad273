#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (build container only).

    python oracle/gen_golden.py            # needs /root/reference; writes tests/golden/*.npz

Fixtures are data only: inputs, every RNG draw the reference consumed, and the reference's
outputs.  Nothing from the reference's sources is stored.
"""
import os
import sys
import warnings
from functools import partial

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

qinfer = ref_shim.install()
from qinfer.tomography import TomographyModel, pauli_basis, gell_mann_basis  # noqa: E402
import np_oracle as orc  # noqa: E402  (only for the restated Ginibre prior; qutip is absent)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


# ------------------------------------------------------------------------------------------
class Recorder:
    """Wraps the global legacy RNG entry points the hot path uses and logs every draw."""

    def __init__(self):
        self.kinds, self.shapes, self.data = [], [], []
        self.enabled = True
        self._random = np.random.random
        self._randn = np.random.randn
        self._normal = np.random.normal

    def random(self, size=None):
        r = self._random(size)
        if self.enabled:
            shape = tuple(np.shape(r))
            self.kinds.append(0)
            self.shapes.append(shape)
            self.data.append(np.asarray(r, dtype=np.float64).ravel().copy())
        return r

    def randn(self, *shape):
        r = self._randn(*shape)
        if self.enabled:
            self.kinds.append(1)
            self.shapes.append(tuple(np.shape(r)))
            self.data.append(np.asarray(r, dtype=np.float64).ravel().copy())
        return r

    def normal(self, loc=0.0, scale=1.0, size=None):
        r = self._normal(loc, scale, size)
        if self.enabled:
            assert loc == 0.0 and scale == 1.0
            self.kinds.append(2)
            self.shapes.append(tuple(np.shape(r)))
            self.data.append(np.asarray(r, dtype=np.float64).ravel().copy())
        return r

    def __enter__(self):
        np.random.random = self.random
        np.random.normal = self.normal
        return self

    def __exit__(self, *a):
        np.random.random = self._random
        np.random.normal = self._normal

    def arrays(self):
        shp = -np.ones((len(self.shapes), max([2] + [len(s_) for s_ in self.shapes])), dtype=np.int64)
        for i, s in enumerate(self.shapes):
            shp[i, :len(s)] = s
        data = np.concatenate(self.data) if self.data else np.zeros(0)
        return dict(draw_kinds=np.array(self.kinds, dtype=np.int8), draw_shapes=shp, draw_data=data)


def run_trajectory(name, model, prior, n_particles, expparams, outcomes_fn, seed=0,
                   batch_interval=None, cov_stride=None, **lw_kwargs):
    """Run reference SMCUpdater over a schedule, recording draws and the per-datum state."""
    np.random.seed(seed)
    rec = Recorder()
    means, covs, esss, rcs, norms = [], [], [], [], []
    outcomes = []
    with rec, warnings.catch_warnings():
        warnings.simplefilter("ignore")
        resampler = qinfer.LiuWestResampler(kernel=rec.randn, default_n_particles=n_particles,
                                            **lw_kwargs)
        upd = qinfer.SMCUpdater(model, n_particles, prior, resampler=resampler)
        x0 = upd.particle_locations.copy()
        n_prior_draws = len(rec.kinds)
        for k in range(expparams.shape[0]):
            ep = expparams[k:k + 1]
            rec.enabled = False
            o = outcomes_fn(k, ep)
            rec.enabled = True
            outcomes.append(o)
            if batch_interval is None:
                upd.update(o, ep)
            else:
                upd.update(o, ep, check_for_resample=False)
                if (k + 1) % batch_interval == 0:
                    upd._maybe_resample()
            means.append(upd.est_mean())
            if cov_stride is None or k % cov_stride == 0:     # (wide clouds: a 64 x 64 covariance per datum is 32 KB)
                covs.append(upd.est_covariance_mtx())
            esss.append(upd.n_ess)
            rcs.append(upd.resample_count)
            norms.append(upd.normalization_record[-1])
    out = dict(seed=seed, n_particles=n_particles, x0=x0, outcomes=np.array(outcomes),
               means=np.array(means), covs=np.array(covs), n_ess=np.array(esss),
               resample_count=np.array(rcs), norms=np.array(norms),
               final_locs=upd.particle_locations, final_weights=upd.particle_weights,
               min_n_ess=upd.min_n_ess, n_prior_draws=n_prior_draws, **rec.arrays())
    if cov_stride is not None:
        out["cov_stride"] = cov_stride                        # covs[j] belongs to datum j * cov_stride
    for fname in (expparams.dtype.names or ()):
        out["ep_" + fname] = expparams[fname]
    if expparams.dtype.names is None:
        out["ep_t"] = expparams
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s resamples=%d mean=%s size=%.0f KB" % (
        name, upd.resample_count, np.array2string(upd.est_mean()[:3], precision=8),
        os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------
def g1_precession():
    m = qinfer.SimplePrecessionModel()
    true = np.array([[0.3]])
    ts = (9 / 8) ** np.arange(200.0)
    sim = lambda k, ep: m.simulate_experiment(true, ep)
    for n in (1000, 256):
        run_trajectory("g1_precession_n%d" % n, m, qinfer.UniformDistribution([0, 1]), n, ts, sim)
    # batch_update semantics (ESS check every 5 data), smc.py:484-487
    run_trajectory("g1_precession_batch5", m, qinfer.UniformDistribution([0, 1]), 512, ts[:100],
                   sim, batch_interval=5, seed=3)


def g1_binomial():
    m = qinfer.BinomialModel(qinfer.SimplePrecessionModel())
    true = np.array([[0.3]])
    ep = np.empty((60,), dtype=m.expparams_dtype)
    ep['x'] = (9 / 8) ** np.arange(60.0)
    ep['n_meas'] = 25
    sim = lambda k, e: int(np.random.binomial(25, np.sin(0.3 * e['x'][0] / 2) ** 2))
    run_trajectory("g1_binomial_n1000", m, qinfer.UniformDistribution([0, 1]), 1000, ep, sim)


def g1_rb():
    m = qinfer.RandomizedBenchmarkingModel()
    prior = qinfer.PostselectedDistribution(
        qinfer.UniformDistribution([[0.8, 1], [0, 1], [0, 1]]), m)
    true = np.array([[0.95, 0.3, 0.5]])
    ep = np.empty((100,), dtype=m.expparams_dtype)
    ep['m'] = 1 + 5 * np.arange(100)
    sim = lambda k, e: m.simulate_experiment(true, e)
    run_trajectory("g1_rb_n2000", m, prior, 2000, ep, sim)


class FixedPrior(qinfer.Distribution):
    def __init__(self, samples):
        self._s = samples

    @property
    def n_rvs(self):
        return self._s.shape[1]

    def sample(self, n=1):
        assert n == self._s.shape[0]
        return self._s.copy()


def g1_tomography():
    basis = pauli_basis(2)
    m = TomographyModel(basis)
    rng = np.random.RandomState(7)
    x0 = orc.ginibre_prior_sample(300, basis.data, rng)
    true = orc.ginibre_prior_sample(1, basis.data, rng)
    ep = np.zeros((40,), dtype=m.expparams_dtype)
    paulis = rng.randint(1, 16, size=40)
    # In the orthonormal Pauli basis B_a = P_a / 2 the projector (I + P)/2 = B_0 + B_p, so its
    # coefficient vector is e_0 + e_p (RandomPauliHeuristic, tomography/expdesign.py:134-160).
    for k, p in enumerate(paulis):
        ep['meas'][k, 0] = 1.0
        ep['meas'][k, p] = 1.0
    sim = lambda k, e: m.simulate_experiment(true, e)
    run_trajectory("g1_tomography_n300", m, FixedPrior(x0), 300, ep, sim)


# ------------------------------------------------------------------------------------------
def g2_likelihoods():
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # precession
        m = qinfer.SimplePrecessionModel()
        omega = np.concatenate([np.linspace(0, 1, 257), [1e-300, 0.3, 0.29999981]])[:, None]
        ts = (9 / 8) ** np.array([0.0, 50, 100, 150, 199])
        out['prec_x'], out['prec_t'] = omega, ts
        out['prec_L'] = m.likelihood(np.array([0, 1]), omega, ts)
        # binomial
        bm = qinfer.BinomialModel(m)
        ep = np.empty((3,), dtype=bm.expparams_dtype)
        ep['x'] = [1.0, (9 / 8) ** 30, (9 / 8) ** 120]
        ep['n_meas'] = [25, 25, 7]
        out['bin_x'], out['bin_t'], out['bin_n'] = omega, ep['x'], ep['n_meas']
        out['bin_L'] = bm.likelihood(np.arange(26), omega, ep)
        # RB
        rb = qinfer.RandomizedBenchmarkingModel()
        g = np.array([0.0, 0.35, 0.9, 1.0])
        P, A, B = np.meshgrid(np.array([0.0, 0.8, 0.95, 0.999, 1.0]), g, g, indexing='ij')
        x = np.stack([P.ravel(), A.ravel(), B.ravel()], axis=1)
        ep = np.empty((5,), dtype=rb.expparams_dtype)
        ep['m'] = [0, 1, 10, 800, 100000]
        out['rb_x'], out['rb_m'] = x, ep['m']
        out['rb_L'] = rb.likelihood(np.array([0, 1]), x, ep)
        out['rb_valid'] = rb.are_models_valid(x + np.array([0, 0.02, -0.01]))
        out['rb_valid_x'] = x + np.array([0, 0.02, -0.01])
        rbi = qinfer.RandomizedBenchmarkingModel(interleaved=True)
        rs = np.random.RandomState(5)
        xi = rs.uniform(0, 1.1, size=(64, 4))
        ep = np.empty((4,), dtype=rbi.expparams_dtype)
        ep['m'] = [1, 7, 50, 400]
        ep['reference'] = [True, False, True, False]
        out['rbi_x'], out['rbi_m'], out['rbi_ref'] = xi, ep['m'], ep['reference']
        out['rbi_L'] = rbi.likelihood(np.array([0, 1]), xi, ep)
        out['rbi_valid'] = rbi.are_models_valid(xi)
        # tomography
        basis = pauli_basis(2)
        tm = TomographyModel(basis)
        rs = np.random.RandomState(11)
        xt = orc.ginibre_prior_sample(64, basis.data, rs)
        xt[::7] *= 1.3                               # some unphysical ones: exercises the clip
        ep = np.zeros((15,), dtype=tm.expparams_dtype)
        for p in range(1, 16):
            ep['meas'][p - 1, 0] = 1.0
            ep['meas'][p - 1, p] = 1.0
        out['tomo_x'], out['tomo_meas'] = xt, ep['meas']
        out['tomo_L'] = tm.likelihood(np.array([0, 1]), xt, ep)
        out['pauli2_basis'] = basis.data
        out['gell_mann3_basis'] = qinfer.tomography.gell_mann_basis(3).data
    np.savez_compressed(os.path.join(OUT, "g2_likelihoods.npz"), **out)
    print("g2_likelihoods")


def g2_edges():
    """Likelihood corners the main G2 grid does not reach: arguments of cos beyond 1e10 rad (the device code leaves
    its in-range argument reduction there), non-finite parameters, and binomial pmfs with many measurements (beyond
    the small-n closed form; beyond the largest finite binomial coefficient)."""
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = qinfer.SimplePrecessionModel()
        omega = np.concatenate([np.linspace(0.01, 1, 64), [0.3, 0.29999981, 1e-300, 0.0, np.inf, np.nan, -0.4]])[:, None]
        ts = np.array([2.5e10, 1e11, 3.7e11, 1e12, 7.7e15, 1e300])
        out['prec_x'], out['prec_t'] = omega, ts
        out['prec_L'] = m.likelihood(np.array([0, 1]), omega, ts)
        bm = qinfer.BinomialModel(m)
        om2 = np.linspace(0.02, 1, 48)[:, None]
        ns = [65, 100, 1000, 1100, 5000]
        ep = np.empty((len(ns),), dtype=bm.expparams_dtype)
        ep['x'] = [1.0, 7.3, 2.1, 3.3, 0.9]
        ep['n_meas'] = ns
        out['bin_x'], out['bin_t'], out['bin_n'] = om2, ep['x'], ep['n_meas']
        ks = np.array([0, 1, 2, 17, 32, 50, 64, 65, 99, 100, 333, 550, 999, 1000, 1099, 1100, 2500, 4999, 5000])
        out['bin_k'] = ks
        # (one experiment at a time: the outcome domain depends on n_meas)
        L = np.zeros((len(ks), om2.shape[0], len(ns)))
        for e in range(len(ns)):
            ok = ks <= ns[e]
            L[ok, :, e] = bm.likelihood(ks[ok], om2, ep[e:e + 1])[:, :, 0]
        out['bin_L'] = L
    np.savez_compressed(os.path.join(OUT, "g2_edges.npz"), **out)
    print("g2_edges")


def g1_clouds():
    """Config 1 once more (same seed, same data as g1_precession_n1000) with the cloud the reference holds after
    EVERY resample: lets a test force the device updater onto the reference's trajectory at each resample and then
    hold it to the reference's final state (the free-running comparison stops at the conditioning horizon)."""
    prior = qinfer.UniformDistribution([0, 1])
    m = qinfer.SimplePrecessionModel()
    true = np.array([[0.3]])
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        upd = qinfer.SMCUpdater(m, 1000, prior)
        x0 = upd.particle_locations.copy()
        clouds, at, outcomes, ts, rcs, means, norms, ess = [], [], [], [], [], [], [], []
        for k in range(200):
            t = np.array([(9 / 8) ** k])
            d = m.simulate_experiment(true, t)
            before = upd.resample_count
            upd.update(d, t)
            if upd.resample_count != before:
                clouds.append(upd.particle_locations.copy())
                at.append(k)
            outcomes.append(int(d))
            ts.append(t[0])
            rcs.append(upd.resample_count)
            means.append(upd.est_mean().copy())
            norms.append(float(np.ravel(upd.normalization_record[-1])[0]))
            ess.append(float(upd.n_ess))
    np.savez_compressed(os.path.join(OUT, "g1_precession_n1000_clouds.npz"), x0=x0, outcomes=np.array(outcomes),
                        ep_t=np.array(ts), resample_count=np.array(rcs), resample_at=np.array(at),
                        clouds=np.array(clouds), means=np.array(means), norms=np.array(norms), n_ess=np.array(ess),
                        final_locs=upd.particle_locations.copy(), final_weights=upd.particle_weights.copy(),
                        final_mean=upd.est_mean(), final_cov=upd.est_covariance_mtx())
    print("g1_precession_n1000_clouds", len(at), "resamples, final mean", upd.est_mean())


def g3_moments():
    out = {}
    rs = np.random.RandomState(21)
    cases = []
    for d, n in [(1, 1), (1, 7), (1, 1000), (1, 65537), (3, 1), (3, 7), (3, 1000), (3, 20011),
                 (16, 1), (16, 7), (16, 1000), (16, 4099)]:
        if True:
            x = rs.randn(n, d) * rs.uniform(0.1, 2, size=d) + rs.uniform(-1, 1, size=d)
            w = rs.random_sample(n) ** 3
            w /= w.sum()
            cases.append(("d%d_n%d" % (d, n), w, x))
    x = 0.3 + 2e-6 * rs.randn(5000, 1)
    w = np.ones(5000) / 5000
    cases.append(("tight", w, x))
    cases.append(("zerocov", np.ones(16) / 16, np.full((16, 2), 0.25)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, w, x in cases:
            pd = qinfer.ParticleDistribution(particle_locations=x, particle_weights=w)
            out[tag + "_w"], out[tag + "_x"] = pd.particle_weights, x
            out[tag + "_mean"] = pd.est_mean()
            cov = pd.est_covariance_mtx()
            out[tag + "_cov"] = cov
            out[tag + "_ess"] = pd.n_ess
            S, err = qinfer.utils.sqrtm_psd(cov)
            out[tag + "_sqrt"], out[tag + "_sqrt_err"] = S, err
    out['tags'] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(OUT, "g3_moments.npz"), **out)
    print("g3_moments")


def g4_liu_west():
    """Single LiuWestResampler calls incl. Q1 (forced invalid particles) and n_particles != N."""
    out = {}
    tags = []

    def one(tag, model, w, x, seed, n_particles=None, **kw):
        np.random.seed(seed)
        rec = Recorder()
        with rec, warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = qinfer.LiuWestResampler(kernel=rec.randn, **kw)
            pd = qinfer.ParticleDistribution(particle_locations=x, particle_weights=w)
            new = res(model, pd, n_particles=n_particles)
        out[tag + "_w"], out[tag + "_x"] = pd.particle_weights, x
        out[tag + "_new"] = new.particle_locations
        for k, v in rec.arrays().items():
            out[tag + "_" + k] = v
        out[tag + "_a"] = kw.get('a', 0.98)
        out[tag + "_h"] = kw.get('h', np.nan) if kw.get('h') is not None else np.nan
        out[tag + "_n_out"] = new.particle_locations.shape[0]
        tags.append(tag)

    rs = np.random.RandomState(31)
    prec = qinfer.SimplePrecessionModel()
    # Q1: 8 particles near zero so several land invalid (SURVEY Appendix B example, seed 1)
    x = np.abs(rs.randn(8, 1)) * 0.3
    w = rs.random_sample(8)
    one("q1_small", prec, w, x, seed=1, a=0.9)
    x = np.abs(0.05 + 0.05 * rs.randn(2000, 1))
    w = rs.random_sample(2000) ** 2
    one("prec_d1", prec, w, x, seed=2)
    one("prec_d1_grow", prec, w, x, seed=3, n_particles=3000)
    one("prec_d1_a1", prec, w, x, seed=4, a=1.0, h=0.005)
    rb = qinfer.RandomizedBenchmarkingModel()
    x = np.stack([rs.uniform(0.9, 1, 1500), rs.uniform(0.2, 0.5, 1500), rs.uniform(0.4, 0.6, 1500)], 1)
    w = rs.random_sample(1500)
    one("rb_d3", rb, w, x, seed=5, a=0.9)
    basis = pauli_basis(2)
    tm = TomographyModel(basis)
    x = orc.ginibre_prior_sample(400, basis.data, rs)
    w = rs.random_sample(400)
    one("tomo_d16", tm, w, x, seed=6)
    out['tags'] = np.array(tags)
    np.savez_compressed(os.path.join(OUT, "g4_liu_west.npz"), **out)
    print("g4_liu_west", tags)


def g5_canonicalize():
    basis = pauli_basis(2)
    tm = TomographyModel(basis)
    rs = np.random.RandomState(41)
    x = orc.ginibre_prior_sample(256, basis.data, rs)
    x[:, 1:] += 0.15 * rs.randn(256, 15)          # many become non-PSD
    x[:, 0] = 0.5 + 0.01 * rs.randn(256)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = tm.canonicalize(x.copy())
        tm2 = TomographyModel(basis, allow_subnormalized=True)
        y2 = tm2.canonicalize(x.copy())
    np.savez_compressed(os.path.join(OUT, "g5_canonicalize.npz"), x=x, y=y, y_subnorm=y2,
                        basis=basis.data)
    print("g5_canonicalize")


def g5_canonicalize_qutrit():
    """tomography/models.py:149-209 on a qutrit (gell_mann_basis(3), d = 9): the dimension between the two BASELINE uses."""
    basis = gell_mann_basis(3)
    tm = TomographyModel(basis)
    rs = np.random.RandomState(43)
    x = orc.ginibre_prior_sample(256, basis.data, rs)
    x[:, 1:] += 0.2 * rs.randn(256, 8)            # many become non-PSD
    x[:, 0] = 1 / np.sqrt(3) + 0.01 * rs.randn(256)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = tm.canonicalize(x.copy())
        tm2 = TomographyModel(basis, allow_subnormalized=True)
        y2 = tm2.canonicalize(x.copy())
    rho = np.tensordot(x, basis.data, 1)
    n_bad = int((np.linalg.eigvalsh(rho).min(axis=1) < 0).sum())
    np.savez_compressed(os.path.join(OUT, "g5_canonicalize_qutrit.npz"), x=x, y=y, y_subnorm=y2, basis=basis.data)
    print("g5_canonicalize_qutrit", n_bad, "of 256 not PSD")


def g6_guards():
    """Mirrors tests/test_smc.py:98-137 on a DecimationModel-style likelihood (0.5 / 0 step)."""
    out = {}

    class Decimation(qinfer.FiniteOutcomeModel):
        n_modelparams = 1
        expparams_dtype = [('alpha', float)]
        is_n_outcomes_constant = True

        def n_outcomes(self, e):
            return 2

        def are_models_valid(self, mp):
            return np.ones(mp.shape[0], dtype=bool)

        def likelihood(self, outcomes, mp, ep):
            pr0 = np.ones((mp.shape[0], 1)) / 2
            pr0[int(np.ceil(ep['alpha'][0] * mp.shape[0])):, :] = 0
            return qinfer.FiniteOutcomeModel.pr0_to_likelihood_array(outcomes, pr0)

    np.random.seed(0)
    n_updates = 6
    N = 4 ** n_updates
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        upd = qinfer.SMCUpdater(Decimation(), N, qinfer.UniformDistribution([0, 1]),
                                resample_thresh=0.0)
        ep = np.empty((1,), dtype=[('alpha', float)])
        mins, esss = [], []
        for k in range(n_updates):
            ep['alpha'][0] = 4.0 ** -(k + 1)
            upd.update(np.array([0]), ep)
            mins.append(upd.min_n_ess)
            esss.append(upd.n_ess)
    out['min_n_ess'], out['n_ess'], out['N'] = np.array(mins), np.array(esss), N
    np.savez_compressed(os.path.join(OUT, "g6_guards.npz"), **out)
    print("g6_guards", mins)


def g7_design():
    """bayes_risk / expected_information_gain of the reference on weighted clouds (smc.py:553-663)."""
    out = {}
    rs = np.random.RandomState(51)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # precession, weighted cloud
        m = qinfer.SimplePrecessionModel()
        np.random.seed(1)
        upd = qinfer.SMCUpdater(m, 3000, qinfer.UniformDistribution([0, 1]))
        for t in (0.7, 1.9, 4.1):
            upd.update(int(rs.randint(2)), np.array([t]))
        ts = np.array([0.5, 2.0, 7.5, 31.0])
        out['prec_x'], out['prec_w'], out['prec_t'] = upd.particle_locations.copy(), upd.particle_weights.copy(), ts
        out['prec_risk'] = upd.bayes_risk(ts)
        out['prec_eig'] = upd.expected_information_gain(ts)
        # binomial (outcome count varies with the experiment)
        bm = qinfer.BinomialModel(qinfer.SimplePrecessionModel())
        np.random.seed(2)
        upd = qinfer.SMCUpdater(bm, 2000, qinfer.UniformDistribution([0, 1]))
        ep = np.empty((3,), dtype=bm.expparams_dtype)
        ep['x'], ep['n_meas'] = [1.0, 3.0, 9.0], [25, 7, 12]
        upd.update(9, ep[0:1])
        out['bin_x'], out['bin_w'] = upd.particle_locations.copy(), upd.particle_weights.copy()
        out['bin_t'], out['bin_n'] = ep['x'], ep['n_meas']
        out['bin_risk'] = upd.bayes_risk(ep)
        out['bin_eig'] = upd.expected_information_gain(ep)
        # RB, d = 3
        rb = qinfer.RandomizedBenchmarkingModel()
        prior = qinfer.PostselectedDistribution(qinfer.UniformDistribution([[0.8, 1], [0, 1], [0, 1]]), rb)
        np.random.seed(3)
        upd = qinfer.SMCUpdater(rb, 2500, prior)
        ep = np.empty((3,), dtype=rb.expparams_dtype)
        ep['m'] = [3, 40, 300]
        upd.update(0, ep[1:2])
        out['rb_x'], out['rb_w'], out['rb_m'] = upd.particle_locations.copy(), upd.particle_weights.copy(), ep['m']
        out['rb_risk'] = upd.bayes_risk(ep)
        out['rb_eig'] = upd.expected_information_gain(ep)
    np.savez_compressed(os.path.join(OUT, "g7_design.npz"), **out)
    print("g7_design", out['prec_risk'], out['bin_eig'], out['rb_risk'])


def g8_binomial_rb():
    """BinomialModel(RandomizedBenchmarkingModel) (the model behind simple_est_rb, simple_est.py:184-254):
    likelihood KATs, one SMC trajectory, and the reference's simple_est_prec / simple_est_rb front-ends on
    fixed data tables."""
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rb = qinfer.RandomizedBenchmarkingModel()
        bm = qinfer.BinomialModel(rb)
        g = np.array([0.0, 0.35, 0.9, 1.0])
        P, A, B = np.meshgrid(np.array([0.0, 0.8, 0.95, 0.999, 1.0]), g, g, indexing='ij')
        x = np.stack([P.ravel(), A.ravel(), B.ravel()], axis=1)
        x = x[rb.are_models_valid(x)]                         # pmf needs 0 <= pr1 <= 1
        ep = np.empty((4,), dtype=bm.expparams_dtype)
        ep['m'] = [0, 1, 10, 800]
        ep['n_meas'] = [25, 25, 7, 40]
        out['brb_x'], out['brb_m'], out['brb_n'] = x, ep['m'], ep['n_meas']
        out['brb_L'] = bm.likelihood(np.arange(41), x, ep)
        rbi = qinfer.RandomizedBenchmarkingModel(interleaved=True)
        bmi = qinfer.BinomialModel(rbi)
        rs = np.random.RandomState(5)
        xi = rs.uniform(0, 1.0, size=(200, 4))
        xi = xi[rbi.are_models_valid(xi)][:48]
        ep = np.empty((4,), dtype=bmi.expparams_dtype)
        ep['m'] = [1, 7, 50, 400]
        ep['reference'] = [True, False, True, False]
        ep['n_meas'] = [10, 25, 25, 3]
        out['brbi_x'], out['brbi_m'], out['brbi_ref'], out['brbi_n'] = xi, ep['m'], ep['reference'], ep['n_meas']
        out['brbi_L'] = bmi.likelihood(np.arange(26), xi, ep)
        out['brbi_dtype_names'] = np.array(list(np.dtype(bmi.expparams_dtype).names))
        out['brb_dtype_names'] = np.array(list(np.dtype(bm.expparams_dtype).names))
    np.savez_compressed(os.path.join(OUT, "g8_binomial_rb.npz"), **out)
    print("g8_binomial_rb", out['brb_L'].shape, out['brbi_L'].shape)
    # trajectory
    prior = qinfer.PostselectedDistribution(qinfer.UniformDistribution([[0.8, 1], [0, 1], [0, 1]]), bm)
    ep = np.empty((60,), dtype=bm.expparams_dtype)
    ep['m'] = 1 + 5 * np.arange(60)
    ep['n_meas'] = 25
    sim = lambda k, e: int(np.random.binomial(25, 0.3 * 0.95 ** int(e['m'][0]) + 0.5))
    run_trajectory("g8_binomial_rb_n1500", bm, prior, 1500, ep, sim)


def g8_simple_est():
    """simple_est_prec / simple_est_rb of the reference (simple_est.py:121-254) on fixed tables under
    np.random.seed(4): a host-RNG (parity-mode) run of the build consumes the same stream."""
    from qinfer.simple_est import simple_est_prec, simple_est_rb
    out = {}
    rs = np.random.RandomState(21)
    # precession table: (counts, t, n_shots)
    ts = np.linspace(1.0, 60.0, 40)
    n_shots = np.full(40, 30)
    counts = rs.binomial(n_shots, np.sin(0.31 * ts / 2) ** 2)
    prec = np.column_stack([counts, ts, n_shots]).astype(float)
    # RB table: (counts, m, n_shots)
    ms = np.arange(1, 160, 4)
    n_shots_rb = np.full(ms.shape, 40)
    counts_rb = rs.binomial(n_shots_rb, 0.3 * 0.97 ** ms + 0.5)
    rbt = np.column_stack([counts_rb, ms, n_shots_rb]).astype(float)
    for name, fn, table, kw in (("prec", simple_est_prec, prec, dict(freq_min=0.0, freq_max=1.0, n_particles=3000)),
                                ("rb", simple_est_rb, rbt, dict(p_min=0.8, p_max=1.0, n_particles=4000))):
        np.random.seed(4)                             # legacy global RNG: prior, resampler u and randn
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mean, cov, extra = fn(table, return_all=True, **kw)
        upd = extra['updater']
        out[name + "_table"] = table
        out[name + "_mean"], out[name + "_cov"] = np.atleast_1d(mean), np.atleast_2d(cov)
        out[name + "_resample_count"] = upd.resample_count
        out[name + "_n_ess"] = upd.n_ess
        print("g8_simple_est", name, mean, upd.resample_count)
    np.savez_compressed(os.path.join(OUT, "g8_simple_est.npz"), **out)


def g9_t2_mle():
    """UnknownT2Model (test_models.py:222-259) and MLEModel (derived_models.py:673-691): likelihood KATs and
    one SMC trajectory each."""
    out = {}
    t2 = qinfer.UnknownT2Model()
    rs = np.random.RandomState(9)
    x = np.column_stack([rs.uniform(0, 1, 96), rs.uniform(0, 0.2, 96)])
    x[:4] = [[0, 0], [1, 0], [0.3, 0.05], [0.5, 1e-300]]
    ep = np.empty((5,), dtype=t2.expparams_dtype)
    ep['t'] = [0.0, 1.0, 17.5, (9 / 8) ** 40, (9 / 8) ** 120]
    out['t2_x'], out['t2_t'] = x, ep['t']
    out['t2_L'] = t2.likelihood(np.array([0, 1]), x, ep)
    out['t2_valid_x'] = np.array([[0.1, 0.1], [-0.1, 0.1], [0.1, -1e-9], [0.0, 0.0]])
    out['t2_valid'] = t2.are_models_valid(out['t2_valid_x'])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, base, gamma in (("mle_prec", qinfer.SimplePrecessionModel(), 2.5),
                                 ("mle_bin", qinfer.BinomialModel(qinfer.SimplePrecessionModel()), 0.5)):
            m = qinfer.MLEModel(base, gamma)
            omega = np.concatenate([np.linspace(0, 1, 65), [0.3, 0.29999981]])[:, None]
            if tag == "mle_prec":
                e = (9 / 8) ** np.array([0.0, 50, 100, 150])
                L = m.likelihood(np.array([0, 1]), omega, e)
                out[tag + '_t'] = e
            else:
                e = np.empty((3,), dtype=base.expparams_dtype)
                e['x'] = [1.0, (9 / 8) ** 30, (9 / 8) ** 90]
                e['n_meas'] = [25, 25, 7]
                L = m.likelihood(np.arange(26), omega, e)
                out[tag + '_t'], out[tag + '_n'] = e['x'], e['n_meas']
            out[tag + '_x'], out[tag + '_gamma'], out[tag + '_L'] = omega, gamma, L
    np.savez_compressed(os.path.join(OUT, "g9_t2_mle.npz"), **out)
    print("g9_t2_mle", out['t2_L'].shape)
    # trajectories
    true = np.array([[0.3, 0.02]])
    ep = np.empty((80,), dtype=t2.expparams_dtype)
    ep['t'] = (9 / 8) ** (np.arange(80.0) * 0.5)
    prior = qinfer.UniformDistribution([[0, 1], [0, 0.1]])
    sim = lambda k, e: t2.simulate_experiment(true, e)
    run_trajectory("g9_unknown_t2_n2000", t2, prior, 2000, ep, sim)
    m = qinfer.MLEModel(qinfer.SimplePrecessionModel(), 3.0)
    ts = (9 / 8) ** np.arange(70.0)
    base = qinfer.SimplePrecessionModel()
    sim = lambda k, e: base.simulate_experiment(np.array([[0.3]]), e)
    run_trajectory("g9_mle_precession_n1000", m, qinfer.UniformDistribution([0, 1]), 1000, ts, sim)


def g10_random_walk():
    """GaussianRandomWalkModel with a fixed diagonal covariance (derived_models.py:743-963): a drifting
    precession frequency tracked through the time-step update (smc.py:447-449); every np.random.normal
    draw of update_timestep is recorded like the resampler's."""
    base = qinfer.SimplePrecessionModel()
    m = qinfer.GaussianRandomWalkModel(base, fixed_covariance=np.array([2.5e-7]))
    ts = np.tile(np.array([4.0, 9.0, 17.0, 30.0, 55.0]), 12)
    rs = np.random.RandomState(8)
    drift = 0.3 + np.cumsum(5e-4 * rs.randn(len(ts)))
    sim = lambda k, e: base.simulate_experiment(np.array([[drift[k]]]), e)
    run_trajectory("g10_grw_precession_n800", m, qinfer.UniformDistribution([0, 1]), 800, ts, sim)
    # two parameters, only the first one walks; experiment-dependent step scale
    t2 = qinfer.UnknownT2Model()
    m2 = qinfer.GaussianRandomWalkModel(t2, random_walk_idxs=[0], fixed_covariance=np.array([1e-6]),
                                        scale_mult=lambda ep: np.sqrt(ep['t']))
    ep = np.empty((48,), dtype=t2.expparams_dtype)
    ep['t'] = np.tile(np.array([2.0, 5.0, 11.0, 23.0]), 12)
    sim2 = lambda k, e: t2.simulate_experiment(np.array([[0.3, 0.02]]), e)
    run_trajectory("g10_grw_t2_n800", m2, qinfer.UniformDistribution([[0, 1], [0, 0.1]]), 800, ep, sim2)


def g10_walk_steps():
    """One seeded `GaussianRandomWalkModel.update_timestep` call of the reference for each covariance variant
    (derived_models.py:920-963: given diagonal, given dense, learned diagonal, learned dense): the cloud in, the global
    seed, the stepped cloud out -- pins the draw SHAPE and ORDER of every variant (the trajectories above pin the diagonal
    one only)."""
    out = {}
    rs = np.random.RandomState(19)
    t2 = qinfer.UnknownT2Model()
    ep = np.empty((3,), dtype=t2.expparams_dtype)
    ep['t'] = [2.0, 5.0, 11.0]
    n = 7
    base = np.column_stack([rs.uniform(0.2, 0.8, n), rs.uniform(0.01, 0.09, n)])
    cov = np.array([[4e-4, 1e-4], [1e-4, 9e-4]])
    variants = {
        "known_diag": (dict(fixed_covariance=np.array([4e-4, 9e-4])), base),
        "known_dense": (dict(fixed_covariance=cov, diagonal=False), base),
        "learned_diag": (dict(), np.column_stack([base, rs.uniform(0.0, 0.03, (n, 2))])),
        "learned_dense": (dict(diagonal=False), np.column_stack([base, rs.uniform(0.0, 0.03, (n, 3))])),
    }
    out["expparam_t"] = ep['t'].astype(np.float64)
    out["cov_dense"] = cov
    out["tags"] = np.array(sorted(variants))
    for tag, (kw, mp) in variants.items():
        m = qinfer.GaussianRandomWalkModel(t2, scale_mult=lambda e: np.sqrt(e['t']), **kw)
        assert m.n_modelparams == mp.shape[1], (tag, m.n_modelparams)
        np.random.seed(4242)
        out[tag + "_in"] = mp
        out[tag + "_out"] = np.asarray(m.update_timestep(mp.copy(), ep), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g10_grw_steps.npz"), **out)


def g11_readouts():
    """Posterior read-outs of a weighted cloud: est_entropy (distributions.py:457-464),
    est_credible_region (:558-614), sample (:320-333, its uniforms recorded) and SMCUpdater.posterior_marginal
    (smc.py:672-716), all computed by the reference on a fixed cloud with distinct weights."""
    out = {}
    rs = np.random.RandomState(31)
    n, d = 5000, 2
    x = np.column_stack([0.3 + 0.02 * rs.randn(n), rs.uniform(0, 0.1, n)])
    w = rs.random_sample(n) ** 3
    w[rs.choice(n, 40, replace=False)] = 0.0            # some exactly-zero weights (entropy skips them)
    w /= w.sum()
    pd = qinfer.ParticleDistribution(particle_locations=x.copy(), particle_weights=w.copy())
    out['x'], out['w'] = x, pd.particle_weights
    out['entropy'] = pd.est_entropy()
    for lvl in (0.5, 0.95):
        inside, outside = pd.est_credible_region(level=lvl, return_outside=True)
        out['cred_%d_inside' % int(lvl * 100)] = inside
        out['cred_%d_n_outside' % int(lvl * 100)] = outside.shape[0]
    out['cred_95_slice0'] = pd.est_credible_region(level=0.95, modelparam_slice=slice(0, 1))
    np.random.seed(12)
    u = np.random.random((64,))
    np.random.seed(12)
    out['sample_u'], out['sample'] = u, pd.sample(n=64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        upd = qinfer.SMCUpdater(qinfer.UnknownT2Model(), n, FixedPrior(x.copy()))
        upd.particle_weights[:] = pd.particle_weights
        ps, pr = upd.posterior_marginal(idx_param=0, res=60)
        out['marg0_ps'], out['marg0_pr'] = ps, pr
        ps, pr = upd.posterior_marginal(idx_param=1, res=40, smoothing=0.004, range_min=0.0, range_max=0.1)
        out['marg1_ps'], out['marg1_pr'] = ps, pr
    np.savez_compressed(os.path.join(OUT, "g11_readouts.npz"), **out)
    print("g11_readouts", out['entropy'], out['cred_95_inside'].shape, out['cred_50_inside'].shape)


def g12_perf_test():
    """perf_test (perf_testing.py:182-295) of the reference under np.random.seed(6): the record array of
    one trial with the t_k = (9/8)^k heuristic -- the loop that defines the benchmark's metric."""
    from qinfer.perf_testing import perf_test
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.random.seed(6)
        perf = perf_test(qinfer.SimplePrecessionModel(), 2000, qinfer.UniformDistribution([0, 1]), 60,
                         qinfer.ExpSparseHeuristic)
        for f in ('loss', 'resample_count', 'outcome', 'true', 'est', 'experiment'):
            out['prec_' + f] = perf[f]
        out['prec_fields'] = np.array(perf.dtype.names)
        m = qinfer.UnknownT2Model()
        np.random.seed(7)
        heur = partial(qinfer.ExpSparseHeuristic, t_field='t', other_fields={})
        perf = perf_test(m, 3000, qinfer.UniformDistribution([[0, 1], [0, 0.1]]), 40, heur)
        for f in ('loss', 'resample_count', 'outcome', 'true', 'est', 't'):
            out['t2_' + f] = perf[f]
        out['t2_fields'] = np.array(perf.dtype.names)
    np.savez_compressed(os.path.join(OUT, "g12_perf_test.npz"), **out)
    print("g12_perf_test", out['prec_loss'][-1], out['t2_loss'][-1], out['prec_resample_count'][-1])


def g13_regions():
    """Region estimators built on est_credible_region (distributions.py:616-754; utils.mvee :314-353): convex hull,
    minimum-volume enclosing ellipsoid and the three in_credible_region methods, on a weighted 2-D cloud."""
    out = {}
    rs = np.random.RandomState(44)
    n = 3000
    x = rs.multivariate_normal([2.0, 3.0], [[1.0, 0.4], [0.4, 0.5]], size=n)
    w = rs.random_sample(n) ** 2
    w /= w.sum()
    pd = qinfer.ParticleDistribution(particle_locations=x.copy(), particle_weights=w.copy())
    out['x'], out['w'] = x, pd.particle_weights
    faces, vertices = pd.region_est_hull(level=0.8)
    out['hull_faces_shape'] = np.array(faces.shape)
    out['hull_vertices'] = vertices
    A, c = pd.region_est_ellipsoid(level=0.8, tol=1e-4)
    out['mvee_A'], out['mvee_c'] = A, c
    pts = rs.multivariate_normal([2.0, 3.0], [[2.0, 0.0], [0.0, 2.0]], size=400)
    out['pts'] = pts
    for method in ('pce', 'hpd-hull', 'hpd-mvee'):
        out['in_' + method.replace('-', '_')] = pd.in_credible_region(pts, level=0.8, method=method)
    pts3 = rs.randn(50, 3)
    A3, c3 = qinfer.utils.mvee(pts3, 1e-5)
    out['mvee3_pts'], out['mvee3_A'], out['mvee3_c'] = pts3, A3, c3
    np.savez_compressed(os.path.join(OUT, "g13_regions.npz"), **out)
    print("g13_regions", vertices.shape, [int(out[k].sum()) for k in ('in_pce', 'in_hpd_hull', 'in_hpd_mvee')])


def g14_kl_divergence():
    """Kernel-density KL divergence between particle clouds (distributions.py:466-500, metrics.py:72-106) and the
    per-resample divergences an updater records with track_resampling_divergence=True (smc.py:506-542):
    computed by the reference on fixed clouds, and along a seeded precession / RB run (every RNG draw of the resampler
    is the legacy stream, so a replaying updater reaches the same clouds)."""
    out = {}
    rs = np.random.RandomState(77)
    for tag, n, m, d in (("d1", 400, 300, 1), ("d3", 250, 350, 3)):
        x = 0.3 + 0.01 * rs.randn(n, d)
        y = 0.3 + 0.012 * rs.randn(m, d) + 0.002
        w = rs.random_sample(n) ** 2
        w[rs.choice(n, 10, replace=False)] = 0.0               # est_entropy skips zero weights, the KDE term keeps them
        w /= w.sum()
        v = rs.random_sample(m)
        v /= v.sum()
        p = qinfer.ParticleDistribution(particle_locations=x.copy(), particle_weights=w.copy())
        q = qinfer.ParticleDistribution(particle_locations=y.copy(), particle_weights=v.copy())
        out[tag + '_x'], out[tag + '_w'], out[tag + '_y'], out[tag + '_v'] = x, p.particle_weights, y, q.particle_weights
        out[tag + '_kl'] = p.est_kl_divergence(q)
        out[tag + '_kl_delta'] = p.est_kl_divergence(q, delta=0.05)
    # an updater's own divergences: every call of _kl_divergence from SMCUpdater.resample is recorded with the clouds it
    # compared (the model's scale matrix Q enters through rescaled_distance_mtx; the reference's models all have
    # Q = 1, so one with a non-trivial Q is made here by instance attribute)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, model, n, n_exp, true in (
                ("prec", qinfer.SimplePrecessionModel(), 300, 40, np.array([[0.3]])),
                ("rb", qinfer.RandomizedBenchmarkingModel(), 400, 30, np.array([[0.95, 0.3, 0.5]]))):
            if tag == "rb":
                model._Q = np.array([4.0, 1.0, 0.25])
                prior = qinfer.PostselectedDistribution(qinfer.UniformDistribution([[0.8, 1], [0, 1], [0, 1]]), model)
                eps = [np.array([(1 + 5 * k,)], dtype=model.expparams_dtype) for k in range(n_exp)]
            else:
                prior = qinfer.UniformDistribution([0, 1])
                eps = [np.array([(9 / 8) ** k]) for k in range(n_exp)]
            np.random.seed(21)
            upd = qinfer.SMCUpdater(model, n, prior, track_resampling_divergence=True)
            recs = []
            orig = upd._kl_divergence

            def rec_kl(old_locs, old_w, *a, **k):
                val = orig(old_locs, old_w, *a, **k)
                recs.append((old_locs.copy(), old_w.copy(), upd.particle_locations.copy(),
                             upd.particle_weights.copy(), val))
                return val
            upd._kl_divergence = rec_kl
            for ep in eps:
                upd.update(int(np.ravel(model.simulate_experiment(true, ep))[0]), ep)
            out[tag + '_divergences'] = np.array(upd.resampling_divergences)
            out[tag + '_Q'] = np.asarray(model.Q, dtype=np.float64)
            out[tag + '_n_recorded'] = min(3, len(recs))
            for i, (xo, wo, xn, wn, val) in enumerate(recs[:3]):
                out['%s_r%d_old_x' % (tag, i)], out['%s_r%d_old_w' % (tag, i)] = xo, wo
                out['%s_r%d_new_x' % (tag, i)], out['%s_r%d_new_w' % (tag, i)] = xn, wn
                out['%s_r%d_kl' % (tag, i)] = val
    np.savez_compressed(os.path.join(OUT, "g14_kl_divergence.npz"), **out)
    print("g14_kl_divergence", out['d1_kl'], out['d3_kl'], out['prec_divergences'][:3], out['rb_divergences'][:3])



# ------------------------------------------------------------------------------------------
# Round 6: tomography beyond two qubits (dim 5 .. 8, d = dim^2 up to 64: the wide kernels, csrc/kernels/wide.hpp).  The
# reference handles any dim (tomography/models.py:82-226); these fixtures pin likelihood, moments, canonicalize and a whole
# SMCUpdater trajectory of a three-qubit model.
def _pauli_meas(m, paulis):
    ep = np.zeros((len(paulis),), dtype=m.expparams_dtype)
    for k, p in enumerate(paulis):
        ep['meas'][k, 0] = 1.0            # (I + P) / 2 = B_0 + B_p up to the basis' normalisation: e_0 + e_p
        ep['meas'][k, p] = 1.0
    return ep


def g1_tomography_3q():
    basis = pauli_basis(3)
    m = TomographyModel(basis)
    rng = np.random.RandomState(17)
    x0 = orc.ginibre_prior_sample(200, basis.data, rng)
    true = orc.ginibre_prior_sample(1, basis.data, rng)
    ep = _pauli_meas(m, rng.randint(1, 64, size=240))
    sim = lambda k, e: m.simulate_experiment(true, e)
    run_trajectory("g1_tomography_3q_n200", m, FixedPrior(x0), 200, ep, sim, cov_stride=40)


def g2_tomography_wide():
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, basis in (("3q", pauli_basis(3)), ("gm5", gell_mann_basis(5)),
                           ("q2xq3", qinfer.tomography.tensor_product_basis(gell_mann_basis(2), gell_mann_basis(3)))):
            tm = TomographyModel(basis)
            d = basis.data.shape[0]
            rs = np.random.RandomState(100 + d)
            xt = orc.ginibre_prior_sample(48, basis.data, rs)
            xt[::7] *= 1.3                               # some unphysical ones: exercises the clip
            ep = np.zeros((8,), dtype=tm.expparams_dtype)
            for k in range(5):                            # sparse: e_0 + e_p
                ep['meas'][k, 0] = 1.0
                ep['meas'][k, rs.randint(1, d)] = 1.0
            for k in range(5, 8):                         # dense: a random pure-state projector's coefficients
                v = rs.randn(basis.data.shape[1]) + 1j * rs.randn(basis.data.shape[1])
                v /= np.linalg.norm(v)
                ep['meas'][k] = np.real(np.einsum('aij,ij->a', basis.data.conj(), np.outer(v, v.conj())))
            out[tag + '_x'], out[tag + '_meas'] = xt, ep['meas']
            out[tag + '_L'] = tm.likelihood(np.array([0, 1]), xt, ep)
            out[tag + '_basis'] = basis.data
    np.savez_compressed(os.path.join(OUT, "g2_tomography_wide.npz"), **out)
    print("g2_tomography_wide")


def g3_moments_wide():
    out = {}
    rs = np.random.RandomState(23)
    cases = []
    for d, n in [(17, 33), (25, 300), (36, 1), (49, 257), (64, 7), (64, 520)]:
        x = rs.randn(n, d) * rs.uniform(0.1, 2, size=d) + rs.uniform(-1, 1, size=d)
        w = rs.random_sample(n) ** 3
        w /= w.sum()
        cases.append(("d%d_n%d" % (d, n), w, x))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, w, x in cases:
            pd = qinfer.ParticleDistribution(particle_locations=x, particle_weights=w)
            out[tag + "_w"], out[tag + "_x"] = pd.particle_weights, x
            out[tag + "_mean"] = pd.est_mean()
            cov = pd.est_covariance_mtx()
            out[tag + "_cov"] = cov
            out[tag + "_ess"] = pd.n_ess
            S, err = qinfer.utils.sqrtm_psd(cov)
            out[tag + "_sqrt"], out[tag + "_sqrt_err"] = S, err
    out['tags'] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(OUT, "g3_moments_wide.npz"), **out)
    print("g3_moments_wide")


def g5_canonicalize_wide():
    """tomography/models.py:149-209 at dim 5, 6, 7, 8 (three qubits)."""
    out = {}
    bases = (("gm5", gell_mann_basis(5), 64), ("q2xq3", qinfer.tomography.tensor_product_basis(gell_mann_basis(2), gell_mann_basis(3)), 64),
             ("gm7", gell_mann_basis(7), 48), ("3q", pauli_basis(3), 128))
    for tag, basis, n in bases:
        dim = basis.data.shape[1]
        d = dim * dim
        tm = TomographyModel(basis)
        rs = np.random.RandomState(50 + dim)
        x = orc.ginibre_prior_sample(n, basis.data, rs)
        x[:, 1:] += (0.2 / dim) * rs.randn(n, d - 1) * (rs.random_sample((n, 1)) < 0.7)   # ~70 % pushed out of the cone
        x[:, 0] = 1 / np.sqrt(dim) + 0.01 * rs.randn(n)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = tm.canonicalize(x.copy())
            y2 = TomographyModel(basis, allow_subnormalized=True).canonicalize(x.copy())
        rho = np.tensordot(x, basis.data, 1)
        n_bad = int((np.linalg.eigvalsh(rho).min(axis=1) < 0).sum())
        out[tag + '_x'], out[tag + '_y'], out[tag + '_y_subnorm'], out[tag + '_basis'] = x, y, y2, basis.data
        print("g5_canonicalize_wide", tag, n_bad, "of", n, "not PSD")
    np.savez_compressed(os.path.join(OUT, "g5_canonicalize_wide.npz"), **out)

if __name__ == "__main__":
    g1_precession()
    g1_binomial()
    g1_rb()
    g1_tomography()
    g2_likelihoods()
    g3_moments()
    g4_liu_west()
    g5_canonicalize()
    g5_canonicalize_qutrit()
    g6_guards()
    g7_design()
    g8_binomial_rb()
    g8_simple_est()
    g9_t2_mle()
    g10_random_walk()
    g10_walk_steps()
    g11_readouts()
    g12_perf_test()
    g13_regions()
    g2_edges()
    g1_clouds()
    g14_kl_divergence()
    g1_tomography_3q()
    g2_tomography_wide()
    g3_moments_wide()
    g5_canonicalize_wide()
    print("total bytes:", sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)))
