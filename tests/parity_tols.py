"""Stated fp64 parity tolerances, shared by the oracle-vs-golden and
HIP-vs-oracle tests.

Why trajectory tolerances carry a conditioning factor: the likelihood cos^2(t x / 2) has
dL/dx = O(t), so a 1-ulp difference in a resampled location (e.g. from a different summation
order in the weighted mean) changes later likelihoods by ~eps * t_k * x.  With t_k = (9/8)^k up to
1.5e10 the *problem* amplifies rounding noise to ~1e-6 relative by k = 199; this is measured
between the reference and a line-by-line NumPy restatement of itself (tests/test_oracle_golden.py),
so it is a property of the path, not of the GPU port.  `cond` below is that factor per datum:
t_k (precession), n_meas * t_k (binomial), m_k (RB: p^m), 1 (tomography).
"""
import numpy as np

EPS = float(np.finfo(np.float64).eps)


def rtol_norm(cond):
    """Per-datum normalisation: linear term = ulp-level location noise amplified by dL/dx;
    quadratic term = the reference's own one-pass covariance cancellation (rel. error
    eps mu^2 / cov with cov ~ 1/t^2) feeding the Liu-West kernel width."""
    return 1e-12 + 256 * EPS * cond + 4 * EPS * cond ** 2


def rtol_ess(cond):
    return 1e-11 + 4 * rtol_norm(cond)


# Free-running trajectories are compared datum by datum only while the path is well conditioned;
# past this horizon (k ~ 105 for t_k = (9/8)^k) two IEEE-correct implementations -- including the
# reference vs. a NumPy restatement of itself -- legitimately decorrelate, and beyond t ~ 2e8 the
# reference's covariance is pure rounding noise ("Covariance has zero norm" warnings in C1).
HORIZON_RTOL = 1e-6


def well_conditioned(cond):
    return rtol_norm(cond) < HORIZON_RTOL


def atol_mean(mean):
    return 1e-12 * max(1.0, float(np.max(np.abs(mean))))


def atol_cov(mean, second_moment_trace, n):
    """Cancellation bound for E[xx^T] - mu mu^T computed in one pass."""
    return 64 * EPS * np.sqrt(n) * (float(np.dot(mean, mean)) + float(second_moment_trace))


def max_js_flips(n):
    """CDF-boundary flips allowed between a sequential cumsum and a parallel scan."""
    return int(np.ceil(n * 1e-9)) + 2


def atol_sqrtm_psd(cov):
    """`sqrtm_psd` of a rank-deficient covariance (tomography: x_0 == 1/2 for every particle) is
    ill conditioned: a true-zero eigenvalue comes out of eigh as +-eps*||cov|| and its square
    root, sqrt(eps ||cov||), is injected along a rounding-noise eigenvector.  Locations produced
    by Liu-West from such a covariance agree only to this level between IEEE-correct
    implementations (measured: 3e-11 between the reference and its NumPy restatement)."""
    return 16 * float(np.sqrt(EPS * np.linalg.norm(np.atleast_2d(cov))))
