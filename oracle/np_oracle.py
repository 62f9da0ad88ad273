"""CPU oracle for the SMC hot path (TEST INFRASTRUCTURE -- never imported by the product).

A NumPy restatement of the algorithm QInfer runs in `SMCUpdater.update / batch_update` plus
`LiuWestResampler`, written from the behaviour described in SURVEY.md section 3/8 with each function
citing the reference lines (relative to /root/reference/src/qinfer/) it follows.  It is the
checker for the HIP kernels: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg
of `bench.py` may import it.

Pinning: `tests/test_oracle_golden.py` checks every function below against golden vectors that
`oracle/gen_golden.py` produced by importing and running the reference itself in the build
container (fixtures under `tests/golden/`), including full seeded C1 trajectories with every RNG
draw recorded.  So: parity PINNED (against outputs of the reference run here).

All arithmetic is float64.  Randomness is injected: every function that consumes random numbers
takes an `rng` object with `random(shape)` and `randn(*shape)`; `LegacyRNG` gives NumPy's global
legacy MT19937 stream (what the reference consumes), `ReplayRNG` replays recorded draws.
"""
from __future__ import annotations

import math
import warnings

import numpy as np
import scipy.linalg as sla

EPS = float(np.spacing(1.0))


class ApproximationWarning(RuntimeWarning):
    """_exceptions.py:72-78."""


class ResamplerWarning(RuntimeWarning):
    """_exceptions.py:66-70."""


class ResamplerError(RuntimeError):
    """_exceptions.py:54-64."""


# ----------------------------------------------------------------------------------------------
# RNG plumbing
# ----------------------------------------------------------------------------------------------
class LegacyRNG:
    """NumPy's global legacy stream -- the reference's only RNG."""

    def random(self, shape):
        return np.random.random(shape)

    def randn(self, *shape):
        return np.random.randn(*shape)

    def normal(self, shape):
        return np.random.normal(size=shape)


class ReplayRNG:
    """Replays a recorded draw log: kinds (0 = uniform, 1 = randn, 2 = np.random.normal), shapes, flat data."""

    def __init__(self, kinds, shapes, data):
        self.kinds = [int(k) for k in kinds]
        self.shapes = [tuple(int(v) for v in s if v >= 0) for s in shapes]
        self.data = np.asarray(data, dtype=np.float64)
        self.pos = 0
        self.off = 0

    def _pop(self, kind, shape):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        assert self.pos < len(self.kinds), "draw log exhausted"
        assert self.kinds[self.pos] == kind, "draw kind mismatch at %d" % self.pos
        assert self.shapes[self.pos] == shape, "draw shape mismatch at %d: log %r, asked %r" % (
            self.pos, self.shapes[self.pos], shape)
        n = int(np.prod(shape)) if len(shape) else 1
        out = self.data[self.off:self.off + n].reshape(shape).copy()
        self.pos += 1
        self.off += n
        return out

    def random(self, shape):
        return self._pop(0, shape)

    def randn(self, *shape):
        return self._pop(1, shape)

    def normal(self, shape):
        return self._pop(2, shape)

    @property
    def exhausted(self):
        return self.pos == len(self.kinds)


# ----------------------------------------------------------------------------------------------
# Likelihoods (a6-a10).  All return L[n_outcomes, n_particles, n_experiments].
# ----------------------------------------------------------------------------------------------
def _two_outcome(outcomes, pr0):
    """abstract_model.py:666-686: outcome 0 -> pr0, anything else -> 1 - pr0."""
    outcomes = np.atleast_1d(np.asarray(outcomes))
    pr1 = 1 - pr0
    return np.stack([pr0 if int(o) == 0 else pr1 for o in outcomes], axis=0)


def lik_precession(outcomes, x, t, w_=0.0):
    """test_models.py:123-143 (via :188-197 with w_ = 0): pr0 = cos(t (omega - w_) / 2)^2."""
    x = np.asarray(x, dtype=np.float64).reshape(len(x), -1)[:, :1]
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))
    w_ = np.broadcast_to(np.asarray(w_, dtype=np.float64), t.shape)
    pr0 = np.cos(t[None, :] * (x - w_[None, :]) / 2) ** 2
    return _two_outcome(outcomes, pr0)


def valid_precession(x, min_freq=0.0):
    """test_models.py:109-110."""
    return np.all(np.asarray(x) > min_freq, axis=1)


def binom_pmf(n, k, p):
    """utils.py:106-111 -> scipy.stats.binom(n, p).pmf(k).

    Third-party arithmetic (SciPy/Boost, not under /root/reference): restated as the closed form
    C(n,k) p^k (1-p)^(n-k) with an exactly-rounded integer binomial coefficient; pinned by G2.
    """
    n = np.asarray(n)
    k = int(k)
    p = np.asarray(p, dtype=np.float64)
    out = np.zeros(np.broadcast(n, p).shape, dtype=np.float64)
    nb = np.broadcast_to(n, out.shape)
    pb = np.broadcast_to(p, out.shape)
    for nv in np.unique(nb):
        nv_i = int(nv)
        sel = nb == nv
        if k < 0 or k > nv_i:
            continue
        pp = pb[sel]
        if nv_i <= 64:
            out[sel] = float(math.comb(nv_i, k)) * pp ** k * (1 - pp) ** (nv_i - k)
        else:
            # many measurements: the powers underflow (and eventually the coefficient overflows) long before their
            # product does -- log space, what SciPy's pmf does internally for every n
            lc = math.lgamma(nv_i + 1.0) - math.lgamma(k + 1.0) - math.lgamma(nv_i - k + 1.0)
            with np.errstate(divide='ignore', invalid='ignore'):
                lp = (k * np.log(pp) if k > 0 else 0.0) + ((nv_i - k) * np.log1p(-pp) if nv_i - k > 0 else 0.0)
            out[sel] = np.exp(lc + lp)
    return out


def lik_binomial_precession(outcomes, x, t, n_meas):
    """derived_models.py:314-329: pr1 = underlying L(outcome 1); L[k] = Binom(n_meas, pr1).pmf(k)."""
    outcomes = np.atleast_1d(np.asarray(outcomes))
    pr1 = lik_precession([1], x, t)[0]                        # (N, n_e)
    n_meas = np.atleast_1d(np.asarray(n_meas))
    return np.stack([binom_pmf(n_meas[None, :], int(k), pr1) for k in outcomes], axis=0)


def lik_rb(outcomes, x, m, reference=None):
    """rb.py:178-195: pr0 = 1 - (A p^m + B); interleaved: p -> where(reference, p, p_tilde p)."""
    x = np.asarray(x, dtype=np.float64)
    m = np.atleast_1d(np.asarray(m)).astype(np.float64)[None, :]
    if x.shape[1] == 4:
        p_tilde, p, A, B = (x[:, i:i + 1] for i in range(4))
        ref = np.atleast_1d(np.asarray(reference, dtype=bool))[None, :]
        p = np.where(ref, p, p_tilde * p)
    else:
        p, A, B = (x[:, i:i + 1] for i in range(3))
    pr0 = 1 - (A * p ** m + B)
    return _two_outcome(outcomes, pr0)


def lik_binomial_rb(outcomes, x, m, n_meas, reference=None):
    """BinomialModel over rb.py:178-195 (derived_models.py:314-329): pr1 = L_RB(outcome 1) = A p^m + B
    (computed as 1 - pr0, like the reference's chain of calls); L[k] = Binom(n_meas, pr1).pmf(k)."""
    outcomes = np.atleast_1d(np.asarray(outcomes))
    pr1 = lik_rb([1], x, m, reference)[0]                         # (N, n_e)
    n_meas = np.atleast_1d(np.asarray(n_meas))
    return np.stack([binom_pmf(n_meas[None, :], int(k), pr1) for k in outcomes], axis=0)


def valid_rb(x):
    """rb.py:149-176."""
    x = np.asarray(x)
    if x.shape[1] == 4:
        pc, p, A, B = x.T
        extra = [0 <= pc, pc <= 1, A * pc + B <= 1]
    else:
        p, A, B = x.T
        extra = []
    conds = [0 <= p, p <= 1, 0 <= A, A <= 1, 0 <= B, B <= 1, A + B <= 1, A * p + B <= 1] + extra
    return np.all(conds, axis=0)


def lik_unknown_t2(outcomes, x, t):
    """test_models.py:247-257: visibility = exp(-t T2_inv); pr0 = vis cos^2(w t / 2) + (1 - vis) / 2."""
    x = np.asarray(x, dtype=np.float64)
    w, T2_inv = x[:, 0:1], x[:, 1:2]
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))[None, :]
    vis = np.exp(-t * T2_inv)
    pr0 = vis * np.cos(w * t / 2) ** 2 + (1 - vis) / 2
    return _two_outcome(outcomes, pr0)


def valid_unknown_t2(x):
    """test_models.py:244-245."""
    return np.all(np.asarray(x) >= 0, axis=1)


def lik_tomography(outcomes, x, meas):
    """tomography/models.py:211-226: pr1 = clip(meas . x, 0, 1); pr0 = 1 - pr1."""
    x = np.asarray(x, dtype=np.float64)
    meas = np.asarray(meas, dtype=np.float64).reshape(-1, x.shape[1])
    pr1 = np.clip(np.einsum('ei,mi->me', meas, x), 0, 1)
    return _two_outcome(outcomes, 1 - pr1)


# ----------------------------------------------------------------------------------------------
# Tomography bases / canonicalize / Ginibre prior
# ----------------------------------------------------------------------------------------------
def gell_mann_data(dim):
    """tomography/bases.py:71-111: identity/sqrt(dim), diagonal, symmetric, antisymmetric."""
    b = np.zeros((dim * dim, dim, dim), dtype=complex)
    b[0] = np.eye(dim) / np.sqrt(dim)
    for r in range(1, dim):
        diag = np.zeros(dim)
        diag[:r] = 1.0
        diag[r] = -r
        b[r] = np.diag(diag) / np.sqrt(r + r * r)
    yoff = dim * (dim - 1) // 2
    for i in range(1, dim):
        for j in range(i):
            idx = (i - 1) * i // 2 + j + dim
            b[idx, i, j] = b[idx, j, i] = 1 / np.sqrt(2)
            b[idx + yoff, i, j] = 1j / np.sqrt(2)
            b[idx + yoff, j, i] = -1j / np.sqrt(2)
    return b


def pauli_data(nq=1):
    """tomography/bases.py:137-154: tensor power of gell_mann(2)[[0, 2, 3, 1]] (I, X, Y, Z)/sqrt2."""
    single = gell_mann_data(2)[[0, 2, 3, 1]]
    out = single
    for _ in range(nq - 1):
        out = np.array([np.kron(a, b) for a in out for b in single])
    return out


def tomo_canonicalize(x, basis, allow_subnormalized=False):
    """tomography/models.py:149-209 (quirk Q3: non-Hermitian eig on rho^T, conj cancels)."""
    x = np.array(x, dtype=np.float64, copy=True)
    dim = basis.shape[1]
    flat = basis.reshape(basis.shape[0], -1)
    for i in range(x.shape[0]):
        arr = np.tensordot(x[i], basis.conj(), 1)
        w, v = np.linalg.eig(arr)
        if not np.all(w >= 0):
            w[w < 0] = 0
            new_arr = np.dot(v * w, v.conj().T)
            x[i] = np.real(np.dot(flat, new_arr.flatten()))
    if not allow_subnormalized:
        x = x / (x[:, 0] * np.sqrt(dim))[:, None]
    return x


def ginibre_prior_sample(n, basis, rng):
    """Restated Ginibre(dim, full rank) prior (tomography/distributions.py:168-196 calls
    qutip.rand_dm_ginibre -- qutip 3.2+, absent: parity UNPINNED for this prior; invariants only).
    X = randn + i randn (dim x dim); rho = X X^dagger / tr; x_a = Re tr(B_a^dagger rho)."""
    dim = basis.shape[1]
    flat = basis.reshape(basis.shape[0], -1)
    out = np.empty((n, dim * dim))
    for i in range(n):
        g = rng.randn(dim, dim) + 1j * rng.randn(dim, dim)
        rho = g @ g.conj().T
        rho /= np.trace(rho).real
        out[i] = np.real(flat.conj() @ rho.flatten())
    return out


# ----------------------------------------------------------------------------------------------
# Weights, ESS, moments (a2, a3, a11-a13)
# ----------------------------------------------------------------------------------------------
def hypothetical_update(w, L):
    """smc.py:353-373 for L[n_o, N, n_e] -> (weights[n_o, n_e, N], norm[n_o, n_e, 1])."""
    Lt = np.transpose(L, (0, 2, 1))
    hyp = w * Lt
    norm = hyp.sum(axis=2)[..., None]
    fixed = norm.copy()
    fixed[np.abs(norm) < EPS] = 1
    return hyp / fixed, norm


def n_ess(w):
    """distributions.py:299-307."""
    return 1 / np.sum(w ** 2)


def particle_mean(w, x):
    """distributions.py:337-348 (twin: utils.py:216-232)."""
    return np.dot(w, x)


def particle_cov(w, x, warn=True):
    """distributions.py:351-399 (twin utils.py:235-287): E[x x^T] - mu mu^T, PSD warning."""
    mu = particle_mean(w, x)
    xs = x.T
    cov = np.einsum('i,mi,ni', w, xs, xs) - np.outer(mu, mu)
    assert np.all(np.isfinite(cov))
    if warn and not np.all(np.linalg.eigvals(cov) >= 0):
        warnings.warn('Numerical error in covariance estimation causing positive semidefinite '
                      'violation.', ApproximationWarning)
    return cov


def sqrtm_psd(A):
    """utils.py:593-607: eigh, clamp w <= 0, (v sqrt(w)) v^H, Frobenius error of S S - A."""
    w, v = sla.eigh(A)
    w = np.where(w <= 0, 0.0, w)
    S = (v * np.sqrt(w)).dot(v.conj().T)
    return S, np.linalg.norm(S @ S - A, 'fro')


def _hyp_all_outcomes(w, x, lik, outcomes, expparams_list):
    """smc.py:577-595: hypothetical weights for every outcome, the last one by complement."""
    L = np.concatenate([lik(outcomes[:-1], x, e) for e in expparams_list], axis=2)     # (n_o-1, N, n_e)
    w_hyp, N = hypothetical_update(w, L)
    Lt = np.transpose(L, (0, 2, 1))
    last = (1 - Lt.sum(axis=0)) * w[np.newaxis, :]
    N = np.concatenate([N[:, :, 0], np.sum(last[np.newaxis, :, :], axis=2)], axis=0)
    last = last / N[-1, :, np.newaxis]
    return np.concatenate([w_hyp, last[np.newaxis, :, :]], axis=0), N


def bayes_risk(w, x, lik, outcomes, expparams_list, Q=None):
    """smc.py:553-611."""
    Q = np.ones(x.shape[1]) if Q is None else Q
    w_hyp, N = _hyp_all_outcomes(w, x, lik, outcomes, expparams_list)
    mu = np.dot(w_hyp, x)
    var = np.sum(w_hyp * np.sum(Q * (x[None, None, :, :] - mu[:, :, None, :]) ** 2, axis=3), axis=2)
    return np.sum(N * var, axis=0)


def expected_information_gain(w, x, lik, outcomes, expparams_list):
    """smc.py:613-663."""
    w_hyp, N = _hyp_all_outcomes(w, x, lik, outcomes, expparams_list)
    with np.errstate(divide='ignore', invalid='ignore'):
        kld = np.sum(w_hyp * np.log(w_hyp / w), axis=2)
    return np.sum(N * kld, axis=0)


# ----------------------------------------------------------------------------------------------
# Liu-West resampler (a14, a15), including quirks Q1 (mus truncation) and Q2 (unclamped search)
# ----------------------------------------------------------------------------------------------
def liu_west(w, x, valid_fn, rng, a=0.98, h=None, maxiter=1000, postselect=True,
             zero_cov_comp=1e-10, n_out=None, mean=None, cov=None, legacy_mus_truncation=True,
             trace=None):
    """resamplers.py:256-392.  Returns (new_locs (n_out, d), uniform weights).

    `trace`, if a dict, receives 'js' (first-round ancestor indices) and 'n_rounds'."""
    if h is None:
        h = np.sqrt(1 - a ** 2)                                   # :248-252
    if mean is None:
        mean = particle_mean(w, x)                                # :266-269
    if cov is None:
        cov = particle_cov(w, x)                                  # :270-273
    N, d = x.shape
    if n_out is None:
        n_out = N
    if np.linalg.norm(cov, 'fro') == 0:                           # :283-294
        warnings.warn("Covariance has zero norm; adding in small covariance in resampler. "
                      "Consider increasing n_particles to improve covariance estimates.",
                      ResamplerWarning)
        cov = zero_cov_comp * np.eye(d)
    S, err = sqrtm_psd(cov)                                       # :295
    if not np.isfinite(err):
        raise ResamplerError("Infinite error in computing the square root of the covariance "
                             "matrix. Check that n_ess is not too small.")
    S = np.real(h * S)                                            # :300
    cdf = np.cumsum(w)                                            # :308
    u = rng.random((n_out,))
    js = cdf.searchsorted(u, side='right')                        # :318-321 (Q2: no clamp)
    if trace is not None:
        trace['js'] = js.copy()
    mus = a * x[js, :] + (1 - a) * mean                           # :325
    centres_all = mus
    new = np.empty((n_out, d))
    idxs = np.arange(n_out)
    rounds = 0
    while idxs.size and rounds < maxiter:                         # :327
        rounds += 1
        z = rng.randn(d, mus.shape[0])                            # param-major (d, k)
        new[idxs, :] = mus + np.dot(S, z).T                       # :332
        cand = new[idxs, :]
        ok = valid_fn(cand) if postselect else np.ones(cand.shape[0], dtype=bool)
        bad = np.logical_not(ok)
        idxs = idxs[bad]
        if legacy_mus_truncation:
            mus = mus[:idxs.size, :]                              # :372  <- quirk Q1
        else:
            mus = centres_all[idxs, :]
    if idxs.size:                                                 # :374-381
        warnings.warn("Liu-West resampling failed to find valid models for {} particles within "
                      "{} iterations.".format(idxs.size, maxiter), ResamplerWarning)
    if trace is not None:
        trace['n_rounds'] = rounds
    return new, np.ones(n_out) / n_out


# ----------------------------------------------------------------------------------------------
# Posterior read-outs
# ----------------------------------------------------------------------------------------------
def est_entropy(w):
    """distributions.py:457-464: -sum over nonzero weights of w log w."""
    nz = w[w > 0]
    return -np.sum(np.log(nz) * nz)


def kl_divergence(x, w, other_x, other_w, Q=1.0, delta=1e-2):
    """Kernel-density estimate of KL(p || q) between two weighted clouds (distributions.py:466-487 with the default
    standard-normal kernel; distances from metrics.py:72-106: ||sqrt(Q) (x_i - y_j)||_2):
        -H(w) - (1 / delta) sum_i w_i log( sum_j v_j phi(d_ij / delta) )."""
    x, other_x = np.asarray(x, dtype=np.float64), np.asarray(other_x, dtype=np.float64)
    diff = np.sqrt(Q) * (x[:, None, :] - other_x[None, :, :])
    dist = np.sqrt(np.sum(diff ** 2, axis=-1)) / delta
    K = np.exp(-0.5 * dist ** 2) / np.sqrt(2.0 * np.pi)
    with np.errstate(divide="ignore"):
        inner = np.log(np.sum(np.asarray(other_w) * K, axis=1))
    return -est_entropy(w) - (1.0 / delta) * np.sum(np.asarray(w) * inner, axis=0)


def est_credible_region(w, x, level=0.95, return_outside=False, modelparam_slice=None):
    """distributions.py:558-614: highest-weight particles first until the mass reaches `level`."""
    mps = x[:, modelparam_slice] if modelparam_slice is not None else x
    order = np.argsort(w)[::-1]
    cum = np.cumsum(w[order])
    cred = cum <= level
    cred[np.sum(cred)] = True
    if return_outside:
        return mps[order][cred], mps[order][np.logical_not(cred)]
    return mps[order][cred]


def sample_cloud(w, x, u):
    """distributions.py:320-333 with the uniforms given: inverse-CDF draws, 'right' side, clamped."""
    cdf = np.cumsum(w)
    return x[np.minimum(cdf.searchsorted(u, side='right'), len(cdf) - 1)]


def posterior_marginal(w, x, idx_param=0, res=100, smoothing=0, range_min=None, range_max=None):
    """smc.py:672-716: derivative of the linearly interpolated marginal CDF on a res-point grid, optional
    Gaussian smoothing (SciPy's interp1d / gaussian_filter1d, as the reference calls them)."""
    import scipy.interpolate
    from scipy.ndimage import gaussian_filter1d
    s = np.argsort(x[:, idx_param])
    locs = x[s, idx_param]
    r_min = np.min(locs) if range_min is None else range_min
    r_max = np.max(locs) if range_max is None else range_max
    ps = np.linspace(r_min, r_max, res)
    interp = scipy.interpolate.interp1d(np.append(locs, r_max + np.abs(r_max - r_min)),
                                        np.append(np.cumsum(w[s]), 1), bounds_error=False, fill_value=0,
                                        assume_sorted=True)
    pr = np.gradient(interp(ps), ps[1] - ps[0])
    if smoothing > 0:
        gaussian_filter1d(pr, res * smoothing / (np.abs(r_max - r_min)), output=pr)
    return ps, pr


# ----------------------------------------------------------------------------------------------
# Stateful driver mirroring SMCUpdater (a1, a4, a5, a16, a17)
# ----------------------------------------------------------------------------------------------
class OracleModel:
    """Bundle of callables describing one of the four hot-path models."""

    def __init__(self, name, d, lik, valid, canon=None, timestep=None):
        self.name, self.d, self.lik, self.valid, self.canon = name, d, lik, valid, canon
        self.timestep = timestep          # (x, expparams, rng) -> x after the step, or None (static parameters)


def precession_model(min_freq=0.0):
    return OracleModel('precession', 1,
                       lambda o, x, e: lik_precession(o, x, e['t']),
                       lambda x: valid_precession(x, min_freq))


def binomial_precession_model(min_freq=0.0):
    return OracleModel('binomial_precession', 1,
                       lambda o, x, e: lik_binomial_precession(o, x, e['t'], e['n_meas']),
                       lambda x: valid_precession(x, min_freq))


def rb_model(interleaved=False):
    return OracleModel('rb', 4 if interleaved else 3,
                       lambda o, x, e: lik_rb(o, x, e['m'], e.get('reference')),
                       valid_rb)


def binomial_rb_model(interleaved=False):
    return OracleModel('binomial_rb', 4 if interleaved else 3,
                       lambda o, x, e: lik_binomial_rb(o, x, e['m'], e['n_meas'], e.get('reference')),
                       valid_rb)


def unknown_t2_model():
    return OracleModel('unknown_t2', 2, lambda o, x, e: lik_unknown_t2(o, x, e['t']), valid_unknown_t2)


def mle_model(base, power):
    """derived_models.py:673-691: L ** power; everything else is the decorated model's."""
    return OracleModel('mle_' + base.name, base.d, lambda o, x, e: base.lik(o, x, e) ** power, base.valid, base.canon)


def gaussian_random_walk_model(base, fixed_std, idxs=None, scale_mult=None):
    """derived_models.py:743-963 with a fixed DIAGONAL covariance and no transformation: after each datum
    x[:, idxs] += fixed_std * scale_mult(e) * normal(size=(n_eps, N, n_rw)) (update_timestep :920-963)."""
    fixed_std = np.atleast_1d(np.asarray(fixed_std, dtype=np.float64))
    idxs = np.arange(base.d) if idxs is None else np.atleast_1d(idxs)

    def step(x, e, rng):
        z = rng.normal((1, x.shape[0], len(idxs)))[0]
        mult = 1.0 if scale_mult is None else float(np.ravel(scale_mult(e))[0])
        out = x.copy()
        out[:, idxs] += mult * (fixed_std * z)
        return out
    return OracleModel('grw_' + base.name, base.d, base.lik, base.valid, base.canon, timestep=step)


def tomography_model(basis, allow_subnormalized=False):
    d = basis.shape[0]
    return OracleModel('tomography', d,
                       lambda o, x, e: lik_tomography(o, x, e['meas']),
                       lambda x: np.ones(x.shape[0], dtype=bool),
                       lambda x: tomo_canonicalize(x, basis, allow_subnormalized))


class OracleSMC:
    """Functional twin of smc.py:97-551 on top of the pieces above.

    `expparams` for update() is a dict of per-experiment field arrays (e.g. {'t': [1.5]}).
    """

    def __init__(self, model, n_particles, prior_sample, rng=None, a=0.98, h=None,
                 resample_thresh=0.5, zero_weight_policy='error', zero_weight_thresh=None,
                 maxiter=1000, postselect=True, canonicalize=True, legacy_mus_truncation=True,
                 default_n_particles='same'):
        self.model = model
        self.rng = rng if rng is not None else LegacyRNG()
        self.prior_sample = prior_sample          # callable n -> (n, d)
        self.a, self.h, self.maxiter, self.postselect = a, h, maxiter, postselect
        self.resample_thresh = resample_thresh
        self.policy = zero_weight_policy
        self.zthresh = 10 * EPS if zero_weight_thresh is None else zero_weight_thresh   # smc.py:171-175
        self.canonicalize = canonicalize
        self.legacy_q1 = legacy_mus_truncation
        # smc.py:156-157: the default resampler is LiuWestResampler(default_n_particles=N)
        self.default_n = n_particles if default_n_particles == 'same' else default_n_particles
        self.resample_count = 0
        self.min_n_ess = n_particles
        self.just_resampled = False
        self.data_record, self.normalization_record = [], []
        self.reset(n_particles)

    def reset(self, n=None):
        """smc.py:281-320."""
        n = self.w.shape[0] if n is None else n
        self.w = np.ones(n) / n
        self.x = np.zeros((n, self.model.d))
        self.x[:, :] = self.prior_sample(n)
        if self.canonicalize and self.model.canon is not None:
            self.x[:, :] = self.model.canon(self.x)

    @property
    def n_particles(self):
        return self.x.shape[0]

    @property
    def n_ess(self):
        return n_ess(self.w)

    def est_mean(self):
        return particle_mean(self.w, self.x)

    def est_covariance_mtx(self):
        return particle_cov(self.w, self.x)

    def update(self, outcome, expparams, check_for_resample=True):
        """smc.py:388-457."""
        self.data_record.append(outcome)
        self.just_resampled = False
        L = self.model.lik(np.atleast_1d(outcome), self.x, expparams)
        weights, norm = hypothetical_update(self.w, L)
        if not np.all(weights >= 0):                                      # :416-418
            warnings.warn("Negative weights occured in particle approximation. Smallest weight "
                          "observed == {}. Clipping weights.".format(np.min(weights)),
                          ApproximationWarning)
            np.clip(weights, 0, 1, out=weights)
        if np.sum(weights) <= self.zthresh:                               # :423-436
            if self.policy == 'ignore':
                pass
            elif self.policy == 'skip':
                return
            elif self.policy == 'warn':
                warnings.warn("All particle weights are zero. This will very likely fail quite "
                              "badly.", ApproximationWarning)
            elif self.policy == 'error':
                raise RuntimeError("All particle weights are zero.")
            elif self.policy == 'reset':
                warnings.warn("All particle weights are zero. Resetting from initial prior.",
                              ApproximationWarning)
                self.reset()
            else:
                raise ValueError("Invalid zero-weight policy {} encountered.".format(self.policy))
        self.w[:] = weights[0, 0, :]                                      # :441
        self.normalization_record.append(norm[0][0])
        if self.model.timestep is not None:                               # :447-449
            self.x = self.model.timestep(self.x, expparams, self.rng)
        if self.n_ess <= self.min_n_ess:                                  # :452-453
            self.min_n_ess = self.n_ess
        if check_for_resample:
            self._maybe_resample()

    def batch_update(self, outcomes, expparams_list, resample_interval=5):
        """smc.py:459-487; expparams_list is a sequence of per-experiment dicts."""
        if len(outcomes) != len(expparams_list):
            raise ValueError("The number of outcomes and experiments must match.")
        for idx, (o, e) in enumerate(zip(outcomes, expparams_list)):
            self.update(o, e, check_for_resample=False)
            if (idx + 1) % resample_interval == 0:
                self._maybe_resample()

    def _maybe_resample(self):
        """smc.py:263-277."""
        ess = self.n_ess
        if ess <= 10:
            warnings.warn("Extremely small n_ess encountered ({}). Resampling is likely to fail. "
                          "Consider adding particles, or resampling more often.".format(ess),
                          ApproximationWarning)
        if ess < self.n_particles * self.resample_thresh:
            self.resample()

    def resample(self):
        """smc.py:491-551."""
        if self.just_resampled:
            warnings.warn("Resampling without additional data; this may not perform as desired.",
                          ResamplerWarning)
        self.just_resampled = True
        self.resample_count += 1
        x, w = liu_west(self.w, self.x, self.model.valid, self.rng, a=self.a, h=self.h,
                        maxiter=self.maxiter, postselect=self.postselect, n_out=self.default_n,
                        legacy_mus_truncation=self.legacy_q1)
        self.x, self.w = x, w
        if self.canonicalize and self.model.canon is not None:
            self.x[:, :] = self.model.canon(self.x)
