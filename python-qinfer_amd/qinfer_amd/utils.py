"""Numeric helpers named as in the reference's `qinfer/utils.py` (the legacy moment functions
:216-287 and `sqrtm_psd` :593-607), evaluated by the same kernels / C-ABI host routine the
updater uses."""
import warnings

import numpy as np

__all__ = ["sqrtm_psd", "particle_meanfn", "particle_covariance_mtx", "binomial_pdf", "outer_product",
           "safe_shape", "mvee", "in_ellipsoid", "uniquify"]


def safe_shape(arr, idx=0, default=1):
    shape = np.shape(arr)
    return shape[idx] if idx < len(shape) else default


def outer_product(vec):
    vec = np.asarray(vec)
    return np.outer(vec, vec) if vec.ndim == 1 else np.dot(vec, vec.T)


def sqrtm_psd(A, est_error=True, check_finite=True):
    """Matrix square root of a PSD matrix with negative eigenvalues truncated; optionally also
    ||S S - A||_F.  Runs `qsmc_sqrtm_psd` (cyclic Jacobi, host) from libqsmc_hip."""
    import ctypes as C
    from . import _native
    A = np.ascontiguousarray(A, dtype=np.float64)
    if check_finite and not np.all(np.isfinite(A)):
        raise ValueError("array must not contain infs or NaNs")
    lib = _native.load()
    d = A.shape[0]
    S = np.empty((d, d))
    err = C.c_double()
    _native.check(None, lib.qsmc_sqrtm_psd(_native.f64_ptr(A), d, 1.0, _native.f64_ptr(S), C.byref(err)),
                  "qsmc_sqrtm_psd")
    return (S, err.value) if est_error else S


def particle_meanfn(weights, locations, fn=None):
    """Deprecated twin of ParticleDistribution.particle_mean / est_meanfn."""
    warnings.warn('particle_meanfn is deprecated, please use distributions.ParticleDistribution',
                  DeprecationWarning)
    from .distributions import ParticleDistribution
    if fn is None:
        return ParticleDistribution.particle_mean(weights, locations)
    vals = np.asarray(fn(locations))
    return np.sum(np.asarray(weights) * vals.transpose([1, 0]), axis=1)


def particle_covariance_mtx(weights, locations):
    """Deprecated twin of ParticleDistribution.particle_covariance_mtx."""
    warnings.warn('particle_covariance_mtx is deprecated, please use distributions.ParticleDistribution',
                  DeprecationWarning)
    from .distributions import ParticleDistribution
    return ParticleDistribution.particle_covariance_mtx(weights, locations)


def binomial_pdf(N, n, p):
    """Binom(N, p).pmf(n) evaluated by the binomial-precession kernel's arithmetic is only
    available through BinomialModel; this helper is the host convenience the reference exposes."""
    from scipy.stats import binom
    return binom(N, p).pmf(n)


def mvee(points, tol=0.001):
    """Minimum-volume enclosing ellipsoid of a point set by Khachiyan's algorithm (reference utils.py:314-353,
    after N. Moshtagh): returns (A, c) with the ellipsoid {x : (x - c)^T A (x - c) <= 1}.

    Same iteration as the reference (weights u on the lifted points Q = [x; 1]; the most outlying point gains
    weight until the step is below `tol`), written without the N x N diagonal matrices."""
    points = np.asarray(points, dtype=np.float64)
    n, d = points.shape
    Q = np.vstack([points.T, np.ones(n)])                 # (d + 1, n)
    u = np.full(n, 1.0 / n)
    err = 1.0
    while err > tol:
        X = (Q * u) @ Q.T
        M = np.einsum('in,ij,jn->n', Q, np.linalg.inv(X), Q)
        j = int(np.argmax(M))
        step = (M[j] - d - 1) / ((d + 1) * (M[j] - 1))
        new_u = (1 - step) * u
        new_u[j] += step
        err = np.linalg.norm(new_u - u)
        u = new_u
    c = points.T @ u
    A = (1.0 / d) * np.linalg.inv((points.T * u) @ points - np.outer(c, c))
    return A, c


def in_ellipsoid(x, A, c):
    """Which of the points x lie in the closed ellipsoid (c - x)^T A^{-1} (c - x) <= 1 (reference utils.py:355-374)."""
    x = np.asarray(x)
    Ainv = np.linalg.inv(A)
    if x.ndim == 1:
        y = c - x
        return np.einsum('j,jl,l', y, Ainv, y) <= 1
    y = c[np.newaxis, :] - x
    return np.einsum('ij,jl,il->i', y, Ainv, y) <= 1


def uniquify(seq):
    """Unique elements of a sequence, first occurrences in order."""
    seen = set()
    return [v for v in seq if not (v in seen or seen.add(v))]
