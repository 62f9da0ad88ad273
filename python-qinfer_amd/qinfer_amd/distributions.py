"""Priors and the device-resident particle cloud.

    Distribution (ABC), UniformDistribution, ProductDistribution, MultivariateNormalDistribution,
    PostselectedDistribution        reference distributions.py:101-127, 753-827, 882-908, 1304-1350
    ParticleDistribution            reference distributions.py:261-464

Priors run once (at `reset`) on the host with NumPy's legacy global RNG -- the same stream the
reference consumes -- or, for a uniform box, optionally on the device with Philox
(`sample_device`).  `ParticleDistribution` is where the data lives: SoA float64 locations
`x[d, N]` and unnormalised weights `w[N]` in HBM, with the normaliser carried as a host scalar.
Its `particle_locations` / `particle_weights` properties materialise NumPy copies in the
reference's layout ((N, d) C-order, normalised weights) only when somebody asks.
"""
import abc
import math
import warnings

import numpy as np

from ._exceptions import ApproximationWarning

__all__ = ["Distribution", "UniformDistribution", "ProductDistribution",
           "MultivariateNormalDistribution", "PostselectedDistribution", "ParticleDistribution"]


class Distribution(metaclass=abc.ABCMeta):
    """A probability distribution over `n_rvs` real random variables."""

    @property
    @abc.abstractmethod
    def n_rvs(self):
        pass

    @abc.abstractmethod
    def sample(self, n=1):
        """(n, n_rvs) array of draws."""


class UniformDistribution(Distribution):
    """Uniform on a box; `ranges` has shape (n_rvs, 2) (or (2,) for one variable)."""

    def __init__(self, ranges=((0, 1),)):
        ranges = np.array(ranges, dtype=float)
        if ranges.ndim == 1:
            ranges = ranges[np.newaxis, ...]
        self._ranges = ranges
        self._n_rvs = ranges.shape[0]
        self._delta = ranges[:, 1] - ranges[:, 0]

    @property
    def n_rvs(self):
        return self._n_rvs

    def sample(self, n=1):
        z = np.random.random((n, self._n_rvs))
        return self._ranges[:, 0] + z * self._delta

    def sample_device(self, engine, n, seed, epoch, model_desc=None, postselect=False, maxiter=100):
        """Philox draw straight into HBM (SoA); used by SMCUpdater(device_rng=True)."""
        from . import _native
        desc = model_desc if model_desc is not None else _native.ModelDesc(
            _native.MODEL_TOMOGRAPHY, self._n_rvs, 0.0, 1, 0)
        return engine.prior_uniform_philox(desc, postselect, self._ranges[:, 0], self._ranges[:, 1],
                                           n, seed, epoch, maxiter)

    def grad_log_pdf(self, var):
        if var.shape[0] == 1:
            return 12 / (self._delta) ** 2
        return np.zeros(var.shape)


class ProductDistribution(Distribution):
    """Cartesian product of independent factor distributions."""

    def __init__(self, *factors):
        if len(factors) == 1 and not isinstance(factors[0], Distribution):
            factors = tuple(factors[0])
        self._factors = list(factors)

    @property
    def n_rvs(self):
        return sum(f.n_rvs for f in self._factors)

    def sample(self, n=1):
        return np.hstack([f.sample(n) for f in self._factors])


class MultivariateNormalDistribution(Distribution):
    def __init__(self, mean, cov):
        import scipy.linalg as la
        self.mean = np.array(mean, dtype=float).flatten()
        self.cov = np.asarray(cov, dtype=float)
        self.invcov = la.inv(self.cov)
        self._sqrt = np.real(la.sqrtm(self.cov))

    @property
    def n_rvs(self):
        return self.mean.shape[0]

    def sample(self, n=1):
        return np.einsum("ij,nj->ni", self._sqrt, np.random.randn(n, self.n_rvs)) + self.mean

    def grad_log_pdf(self, x):
        return -np.dot(self.invcov, (x - self.mean).transpose()).transpose()


class PostselectedDistribution(Distribution):
    """Redraws samples of `distribution` until `model.are_models_valid` accepts them."""

    def __init__(self, distribution, model, maxiters=100):
        self._dist = distribution
        self._model = model
        self._maxiters = maxiters

    @property
    def n_rvs(self):
        return self._dist.n_rvs

    def sample(self, n=1):
        samples = np.empty((n, self.n_rvs))
        todo = np.arange(n)
        iters = 0
        while todo.size and iters < self._maxiters:
            samples[todo] = self._dist.sample(len(todo))
            ok = np.asarray(self._model.are_models_valid(samples[todo, :]), dtype=bool)
            todo = todo[np.logical_not(ok)]
            iters += 1
        if todo.size:
            raise RuntimeError("Did not successfully postselect within {} iterations.".format(self._maxiters))
        return samples

    def sample_device(self, engine, n, seed, epoch, maxiter=None):
        from .abstract_model import native_ok
        if not hasattr(self._dist, "sample_device") or not native_ok(self._model):
            raise NotImplementedError
        x, failed = self._dist.sample_device(engine, n, seed, epoch, self._model._native_desc(), True,
                                             self._maxiters if maxiter is None else maxiter)
        if failed:
            raise RuntimeError("Did not successfully postselect within {} iterations.".format(self._maxiters))
        return x, failed

    def grad_log_pdf(self, x):
        return self._dist.grad_log_pdf(x)


class DeviceBackedArray(np.ndarray):
    """What `particle_locations` / `particle_weights` return: a host snapshot of the device array that WRITES
    THROUGH.  The reference holds these as plain NumPy attributes and mutates them in place
    (`self.particle_weights[:] = ...`, smc.py:441; `self.particle_locations[:, :] = ...`, smc.py:529), so user code
    written against it does the same; here every in-place write -- item / slice assignment, an in-place operator, a
    ufunc with `out=` or `.at`, `fill`, `sort`, `put`, `itemset`-style method calls, `np.copyto` / `np.put` /
    `np.place` / `np.putmask` with the snapshot as destination -- also through a VIEW of the snapshot (a basic slice,
    a transpose) -- is followed by an upload of the whole array to the device.  A snapshot taken before the cloud
    changed underneath it (an update, a resample, another assignment) is stale: reading it gives the old numbers,
    writing to it raises instead of silently losing data.
    Arithmetic on a snapshot, `.copy()`, and fancy / boolean-mask indexing (`snap[mask]`, `snap[[1, 2]]`) return
    arrays that own their memory: editing those touches nothing else, exactly as with the reference's plain arrays.
    One path cannot be seen from here and does NOT write through: writing via `np.asarray(snap)` / a memoryview of the
    snapshot (the base-class view drops the subclass); assign the result back (`upd.particle_weights = w`)."""

    def __new__(cls, arr, owner, what):
        obj = np.asarray(arr).view(cls)
        obj._owner, obj._what, obj._root = owner, what, obj
        obj._version = owner._view_version
        return obj

    def __array_finalize__(self, obj):
        root = getattr(obj, "_root", None)
        # a view INTO a snapshot shares its memory and writes through; anything else derived from one (NumPy wraps the
        # results of fancy / mask indexing and of `take` in the subclass too, with `.base` set to a private buffer) is
        # a copy and behaves like a plain array
        if root is not None and self.base is not None and np.may_share_memory(self, root):
            self._root, self._owner, self._what = root, obj._owner, obj._what
        else:
            self._root = None

    def _push(self):
        root = getattr(self, "_root", None)
        if root is None:
            return
        owner = root._owner
        if owner._view_version != root._version:
            raise RuntimeError("this array is a snapshot of particle_{} taken before the cloud changed (update / "
                               "resample / assignment); take a fresh one before writing".format(root._what))
        owner._write_back(root._what, np.asarray(root))
        root._version = owner._view_version

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        self._push()

    def fill(self, value):
        np.ndarray.fill(self, value)
        self._push()

    def sort(self, *args, **kwargs):
        np.ndarray.sort(self, *args, **kwargs)
        self._push()

    def put(self, *args, **kwargs):
        np.ndarray.put(self, *args, **kwargs)
        self._push()

    def partition(self, *args, **kwargs):
        np.ndarray.partition(self, *args, **kwargs)
        self._push()

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        plain = tuple(np.asarray(i) if isinstance(i, DeviceBackedArray) else i for i in inputs)
        touched = []
        if out is not None:
            touched = [o for o in out if isinstance(o, DeviceBackedArray)]
            kwargs["out"] = tuple(np.asarray(o) if isinstance(o, DeviceBackedArray) else o for o in out)
        if method == "at" and inputs and isinstance(inputs[0], DeviceBackedArray):
            touched.append(inputs[0])                   # ufunc.at(a, idx[, b]) edits its first operand in place
        res = getattr(ufunc, method)(*plain, **kwargs)
        for o in touched:
            o._push()
        if out is not None and len(out) == 1 and touched:
            return out[0]
        return res

    _INPLACE_FUNCS = {np.copyto: "dst", np.put: "a", np.place: "arr", np.putmask: "a"}

    def __array_function__(self, func, types, args, kwargs):
        dst_name = self._INPLACE_FUNCS.get(func)
        if dst_name is None:
            return super().__array_function__(func, types, args, kwargs)
        dst = args[0] if args else kwargs.get(dst_name)
        strip = lambda a: np.asarray(a) if isinstance(a, DeviceBackedArray) else a     # noqa: E731
        res = func(*[strip(a) for a in args], **{k: strip(v) for k, v in kwargs.items()})
        if isinstance(dst, DeviceBackedArray):
            dst._push()
        return res


class ParticleDistribution(Distribution):
    """A weighted particle cloud resident on the GPU.

    Public constructor signature as the reference's (`n_mps` XOR locations+weights, NumPy in).
    """

    def __init__(self, n_mps=None, particle_locations=None, particle_weights=None):
        from .engine import get_engine
        self._eng = get_engine()
        self._moments_cache = None
        if particle_locations is None or particle_weights is None:
            locs = np.zeros((1, n_mps))
            w = np.ones((1,))
        elif n_mps is None:
            locs = np.asarray(particle_locations, dtype=np.float64)
            w = np.abs(np.asarray(particle_weights, dtype=np.float64))
            w = w / np.sum(w) if w.size else w
        else:
            raise ValueError('Either the dimension of parameter space, `n_mps`, or the particles, '
                             '`particle_locations` and `particle_weights` must be specified.')
        self._set_host(locs, w)

    # ---------------------------------------------------------------- device state plumbing
    @classmethod
    def _from_device(cls, engine, x, w, norm=1.0, sumsq=None):
        self = cls.__new__(cls)
        self._eng = engine
        self._x, self._w, self._norm, self._sumsq = x, w, float(norm), sumsq
        self._w_alt = None
        self._moments_cache = None
        return self

    def _set_host(self, locs, w):
        locs = np.asarray(locs, dtype=np.float64)
        if locs.ndim != 2:
            raise ValueError("particle_locations must have shape (n_particles, n_modelparams)")
        self._x = self._eng.locs_to_soa(locs)
        self._w = self._eng.to_device(np.asarray(w, dtype=np.float64))
        self._w_alt = None
        self._norm = 1.0
        self._sumsq = None
        self._moments_cache = None

    _view_version = 0
    _host_locs = None

    def _invalidate(self, locations=True):
        """Something about the cloud changed.  `locations=False`: only the weights did (an update's commit, a weight
        assignment) -- the host copy of the locations kept for plugin callbacks (`SMCUpdater._host_locations`) stays."""
        self._moments_cache = None
        self._w_token = 0            # the weights are no longer (known to be) the output of a fused update
        self._view_version += 1      # host snapshots handed out so far are stale from here on
        self._step_synced = False    # (SMCUpdater: the qsmc_step_t mirror of the cloud must be refilled)
        if locations:
            self._host_locs = None

    def _write_back(self, what, arr):
        """Upload an edited host snapshot (DeviceBackedArray._push)."""
        if what == "weights":
            self.particle_weights = arr
        else:
            self.particle_locations = arr

    def _weights(self):
        """Explicit unnormalised weights.  `_w is None` encodes an all-ones cloud (uniform weights
        after a reset / resample, true weight 1 / _norm): the fused update consumes that form
        directly (w_in = NULL), everything else materialises it here on first use."""
        if self._w is None:
            self._w = self._eng.empty(self.n_particles)
            self._eng.fill(self._w, 1.0)
        return self._w

    def _scratch_weights(self):
        """Where the next update writes its weights: the one of two pooled buffers that is not `_w` (kept across
        resamples, which leave the weights implicit, so that no allocation sits on the per-datum path)."""
        n = self.n_particles
        if self._w_alt is None or self._w_alt.shape[0] != n:
            pool = getattr(self, "_w_pool", None)
            if pool is None or pool[0].shape[0] != n:
                pool = self._w_pool = (self._eng.empty(n), self._eng.empty(n))
            self._w_alt = pool[1] if pool[0] is self._w else pool[0]
        return self._w_alt

    # ---------------------------------------------------------------- reference attributes
    @property
    def particle_locations(self):
        """(N, d) host snapshot of the cloud (D2H) that writes through: `upd.particle_locations[:, 0] = v` uploads."""
        return DeviceBackedArray(np.ascontiguousarray(self._x.cpu().numpy().T), self, "locations")

    @particle_locations.setter
    def particle_locations(self, locs):
        locs = np.asarray(locs, dtype=np.float64)
        self._x = self._eng.locs_to_soa(locs)
        self._shard_sums = None      # (sharded updater: a changed shard size invalidates the resample plan's inputs)
        self._invalidate()

    @property
    def particle_weights(self):
        """(N,) host snapshot of the normalised weights (D2H) that writes through (`upd.particle_weights[:] = w`,
        smc.py:441, uploads)."""
        if self.n_particles == 0:
            return np.zeros((0,))
        return DeviceBackedArray(self._eng.normalized_weights(self._weights(), self._norm).cpu().numpy(), self, "weights")

    @particle_weights.setter
    def particle_weights(self, w):
        self._w = self._eng.to_device(np.asarray(w, dtype=np.float64))
        self._w_alt = None
        self._norm = 1.0
        self._sumsq = None
        self._shard_sums = None      # (sharded updater: per-rank weight totals must be gathered again)
        self._invalidate(locations=False)

    @property
    def n_particles(self):
        return int(self._x.shape[1])

    @property
    def n_rvs(self):
        return int(self._x.shape[0])

    @property
    def n_ess(self):
        """1 / sum_i w_i^2 of the normalised weights."""
        if self._sumsq is None:
            st = self._eng.weight_stats(self._weights(), self._norm)
            self._sumsq = st.sumsq * self._norm * self._norm     # keep it in unnormalised units
        return self._ess_from(self._sumsq)

    def _ess_from(self, sumsq_unnormalised):
        # (plain floats: this sits on the per-datum path; an np.errstate block costs a microsecond)
        s = float(sumsq_unnormalised)
        n2 = float(self._norm) * float(self._norm)
        if s == 0.0:
            return np.float64(np.inf if n2 > 0 else np.nan)
        return np.float64(n2 / s)

    # ---------------------------------------------------------------- moments
    def _moments(self):
        if self._moments_cache is None:
            s0, s1, s2 = self._eng.moments(self._x, self._weights(), self._norm)
            self._moments_cache = (s0, s1, s2)
        return self._moments_cache

    @staticmethod
    def particle_mean(weights, locations):
        """Weighted mean of host arrays, evaluated on the GPU."""
        from .engine import get_engine
        eng = get_engine()
        x = eng.locs_to_soa(np.asarray(locations, dtype=np.float64))
        w = eng.to_device(np.asarray(weights, dtype=np.float64))
        return eng.moments(x, w, 1.0)[1]

    @classmethod
    def particle_covariance_mtx(cls, weights, locations):
        from .engine import get_engine
        eng = get_engine()
        x = eng.locs_to_soa(np.asarray(locations, dtype=np.float64))
        w = eng.to_device(np.asarray(weights, dtype=np.float64))
        _, s1, s2 = eng.moments(x, w, 1.0)
        return cls._cov_from_sums(s1, s2)

    @staticmethod
    def _cov_from_sums(s1, s2, psd_hint=None):
        if s2.shape == (1, 1):
            # one parameter: the same subtraction on plain floats (this is on the resample path of every d = 1 model)
            c = float(s2[0, 0]) - float(s1[0]) * float(s1[0])
            assert math.isfinite(c)
            if not c >= 0:
                warnings.warn('Numerical error in covariance estimation causing positive semidefinite '
                              'violation.', ApproximationWarning)
            return np.array([[c]])
        cov = s2 - np.outer(s1, s1)                       # E[x x^T] - mu mu^T (distributions.py:386-390)
        assert np.all(np.isfinite(cov))
        # (the reference asks the general solver, `la.eig(cov)[0] >= 0`; cov is exactly symmetric here, so the symmetric
        #  one answers the same question at a third of the time -- 15 us instead of 36-74 at d = 16, on every resample)
        if psd_hint is not None and psd_hint[0] == cov.tobytes():
            # this very matrix went through the library's Jacobi a moment ago (a resample queued by qsmc_step): its smallest
            # eigenvalue is known -- the eigendecomposition here (15 us warm, 100+ us on the cold caches of a resample
            # step) would keep the GPU waiting for the next datum
            psd = psd_hint[1] >= 0
        else:
            psd = (cov[0, 0] >= 0) if cov.shape == (1, 1) else np.linalg.eigvalsh(cov)[0] >= 0
        if not psd:
            warnings.warn('Numerical error in covariance estimation causing positive semidefinite '
                          'violation.', ApproximationWarning)
        return cov

    def est_mean(self):
        return self._moments()[1].copy()

    def est_covariance_mtx(self, corr=False):
        _, s1, s2 = self._moments()
        hint = self.__dict__.pop("_queued_psd", None)     # (SMCUpdater: qsmc_step's Jacobi already has the eigenvalues)
        cov = self._cov_from_sums(s1, s2, hint)
        if corr:
            dstd = np.sqrt(np.diag(cov))
            cov = cov / np.outer(dstd, dstd)
        return cov

    def est_meanfn(self, fn):
        """E[fn(x)] for a host-vectorised fn (plugin slow path: evaluates fn on a host copy)."""
        vals = np.asarray(fn(self.particle_locations))
        return np.einsum('i...,i...', self.particle_weights, vals)

    def _single_cloud_only(self, what):
        if getattr(self, "_comm", None) is not None:
            raise NotImplementedError("{} looks at one cloud; a sharded updater holds only its shard".format(what))

    def est_entropy(self):
        """-sum_i w_i log w_i over the particles of nonzero weight (distributions.py:457-464); one pass."""
        self._single_cloud_only("est_entropy")
        return float(self._eng.weight_entropy(self._w, self.n_particles, self._norm))

    def _kl_scale(self):
        """sqrt(Q) of metrics.rescaled_distance_mtx: the model's scale matrix for an updater, 1 for a bare cloud
        (metrics.py:97)."""
        model = getattr(self, "model", None)
        return 1.0 if model is None else np.sqrt(np.asarray(model.Q, dtype=np.float64))

    def _kl_from_device(self, other_x, other_w, other_norm, kernel=None, delta=1e-2):
        """KL(self || other) with `other` a device cloud: SoA locations (d, m), unnormalised weights (None = all
        ones) and their normaliser."""
        self._single_cloud_only("est_kl_divergence")
        if kernel is not None:
            # a user kernel is a host callable: the reference's own O(n m) evaluation on host copies
            y = np.ascontiguousarray(other_x.cpu().numpy().T)
            v = (np.full(y.shape[0], 1.0 / other_norm) if other_w is None
                 else other_w.cpu().numpy() / other_norm)
            x = self.particle_locations
            diff = self._kl_scale() * (x[:, None, :] - y[None, :, :])
            K = kernel(np.sqrt(np.sum(diff ** 2, axis=-1)) / delta)
            with np.errstate(divide="ignore"):
                inner = np.log(np.sum(v * K, axis=1))
            return -self.est_entropy() - (1 / delta) * np.sum(self.particle_weights * inner, axis=0)
        from . import _native
        if self._x.shape[0] > _native.QSMC_MAX_D:
            # wide clouds (16 < d <= 64): the kernel-density kernel keeps a particle in registers (d <= 16); above that the
            # reference's own evaluation with its default kernel (the standard normal pdf), in blocks of rows on the host
            y = np.ascontiguousarray(other_x.cpu().numpy().T)
            v = (np.full(y.shape[0], 1.0 / other_norm) if other_w is None else other_w.cpu().numpy() / other_norm)
            x, w, sc = self.particle_locations, self.particle_weights, self._kl_scale()
            ys, y2 = y * sc, None
            y2 = np.sum(ys * ys, axis=1)
            total = 0.0
            for i0 in range(0, x.shape[0], 1024):
                xs = x[i0:i0 + 1024] * sc
                r2 = np.maximum(np.sum(xs * xs, axis=1)[:, None] + y2[None, :] - 2.0 * xs @ ys.T, 0.0)
                K = np.exp(-0.5 * r2 / (delta * delta)) / np.sqrt(2 * np.pi)
                with np.errstate(divide="ignore"):
                    total += np.sum(w[i0:i0 + 1024] * np.log(np.sum(v * K, axis=1)))
            return -self.est_entropy() - (1 / delta) * total
        cross = self._eng.kde_cross_entropy(self._x, self._w, self._norm, other_x, other_w, other_norm,
                                            self._kl_scale() / delta)
        return -self.est_entropy() - (1 / delta) * cross

    def _kl_divergence(self, other_locs, other_weights, kernel=None, delta=1e-2):
        """KL divergence of this distribution from another cloud, smoothing the other cloud's particles with a
        kernel density estimator (distributions.py:466-487): host arrays (m, d) and (m,) in, evaluated on the GPU."""
        eng = self._eng
        y = eng.locs_to_soa(np.asarray(other_locs, dtype=np.float64))
        v = eng.to_device(np.ascontiguousarray(other_weights, dtype=np.float64))
        return self._kl_from_device(y, v, 1.0, kernel, delta)

    def est_kl_divergence(self, other, kernel=None, delta=1e-2):
        """KL divergence between this and another particle distribution (distributions.py:489-500)."""
        if isinstance(other, ParticleDistribution) and other._eng is self._eng:
            return self._kl_from_device(other._x, other._w, other._norm, kernel, delta)
        return self._kl_divergence(other.particle_locations, other.particle_weights, kernel, delta)

    # ---------------------------------------------------------------- regions / marginals
    def est_credible_region(self, level=0.95, return_outside=False, modelparam_slice=None):
        """Particles of a credible set of mass >= `level`: highest weight first (distributions.py:558-614).

        Sort (descending, device radix sort), scan, cut: returns the (n_credible, n_mps) host array of the
        particles inside -- and with `return_outside` also the rest -- restricted to `modelparam_slice`."""
        self._single_cloud_only("est_credible_region")
        eng = self._eng
        n = self.n_particles
        ws, order = eng.argsort(self._weights(), descending=True)
        cdf = eng.cumsum(ws, self._norm)
        # entries <= level, plus the one that crosses the level
        k = min(int(eng.searchsorted(cdf, [level], side="right").cpu().numpy()[0]) + 1, n)
        rows = self._x if modelparam_slice is None else self._x[modelparam_slice]
        if rows.dim() == 1:
            rows = rows[None, :]
        rows = rows.contiguous()
        inside = np.ascontiguousarray(eng.gather_rows(rows, order[:k].contiguous()).cpu().numpy().T)
        if return_outside:
            if k == n:
                return inside, np.empty((0, inside.shape[1]))
            return inside, np.ascontiguousarray(eng.gather_rows(rows, order[k:].contiguous()).cpu().numpy().T)
        return inside

    def region_est_hull(self, level=0.95, modelparam_slice=None):
        """Convex hull of the credible particle set (distributions.py:616-642): (faces, vertices) with faces of
        shape (n_face, n_mps, n_mps) and the hull's vertices (n_vertices, n_mps).  The credible set comes from the
        device (`est_credible_region`); the hull itself is host geometry (SciPy / Qhull, as in the reference)."""
        from scipy.spatial import ConvexHull
        from . import utils as u
        points = self.est_credible_region(level=level, modelparam_slice=modelparam_slice)
        hull = ConvexHull(points)
        return points[hull.simplices], points[u.uniquify(hull.vertices.flatten())]

    def region_est_ellipsoid(self, level=0.95, tol=0.0001, modelparam_slice=None):
        """Minimum-volume enclosing ellipsoid of that hull (distributions.py:644-667): (A, c) with A the
        ellipsoid's covariance-like shape matrix, x inside iff (x - c)^T A^{-1} (x - c) <= 1."""
        from . import utils as u
        _, vertices = self.region_est_hull(level=level, modelparam_slice=modelparam_slice)
        return u.mvee(vertices, tol)

    def in_credible_region(self, points, level=0.95, modelparam_slice=None, method='hpd-hull', tol=0.0001):
        """Which of `points` lie in a credible region of the cloud (distributions.py:669-754): 'pce' (posterior
        covariance ellipsoid scaled by the chi-square quantile), 'hpd-hull' (convex hull of the highest-weight
        particles, via a Delaunay triangulation) or 'hpd-mvee' (its minimum-volume enclosing ellipsoid)."""
        import scipy.stats as st
        from scipy.spatial import Delaunay
        from . import utils as u
        points = np.asarray(points)
        if method == 'pce':
            s_ = np.s_[modelparam_slice] if modelparam_slice is not None else np.s_[:]
            A = self.est_covariance_mtx()[s_, s_]
            c = self.est_mean()[s_]
            return u.in_ellipsoid(points, st.chi2.ppf(level, c.size) * A, c)
        if method == 'hpd-mvee':
            tol = 0.0001 if tol is None else tol
            A, c = self.region_est_ellipsoid(level=level, tol=tol, modelparam_slice=modelparam_slice)
            return u.in_ellipsoid(points, np.linalg.inv(A), c)
        if method == 'hpd-hull':
            hull = Delaunay(self.est_credible_region(level=level, modelparam_slice=modelparam_slice))
            return hull.find_simplex(points) >= 0
        raise ValueError("method must be 'pce', 'hpd-hull' or 'hpd-mvee'")

    def posterior_marginal(self, idx_param=0, res=100, smoothing=0, range_min=None, range_max=None):
        """Marginal density of one parameter on a `res`-point grid: derivative of the linearly interpolated
        weighted CDF, optionally Gaussian-smoothed (smc.py:672-716).  The cloud is sorted and scanned on
        the device; only the 2 * res bracketing CDF entries travel to the host."""
        from scipy.ndimage import gaussian_filter1d
        self._single_cloud_only("posterior_marginal")
        eng = self._eng
        n = self.n_particles
        locs, order = eng.argsort(self._x[idx_param].contiguous())
        cdf = eng.cumsum(eng.gather_rows(self._weights()[None, :], order)[0], self._norm)
        ends = eng.gather_rows(locs[None, :], eng.to_device(np.array([0, n - 1], dtype=np.int64)))[0].cpu().numpy()
        r_min = float(ends[0]) if range_min is None else range_min
        r_max = float(ends[1]) if range_max is None else range_max
        ps = np.linspace(r_min, r_max, res)
        # piecewise-linear interpolation through (locs, cdf) plus the closing point (r_max + |r_max - r_min|, 1);
        # zero outside [locs[0], closing point]  (interp1d(..., bounds_error=False, fill_value=0))
        x_end = r_max + abs(r_max - r_min)
        hi = np.clip(eng.searchsorted(locs, ps, side="left").cpu().numpy(), 1, n)   # bracket [hi - 1, hi], extended table
        lo = hi - 1
        last = hi == n
        pick = eng.to_device(np.concatenate([lo, np.minimum(hi, n - 1)]).astype(np.int64))
        xs = eng.gather_rows(locs[None, :], pick)[0].cpu().numpy()
        ys = eng.gather_rows(cdf[None, :], pick)[0].cpu().numpy()
        x_lo, y_lo = xs[:res], ys[:res]
        x_hi = np.where(last, x_end, xs[res:])
        y_hi = np.where(last, 1.0, ys[res:])
        with np.errstate(divide='ignore', invalid='ignore'):
            y = (y_hi - y_lo) / (x_hi - x_lo) * (ps - x_lo) + y_lo
        y[(ps < float(ends[0])) | (ps > x_end)] = 0.0
        pr = np.gradient(y, ps[1] - ps[0])
        if smoothing > 0:
            gaussian_filter1d(pr, res * smoothing / abs(r_max - r_min), output=pr)
        return ps, pr

    # ---------------------------------------------------------------- sampling
    def sample(self, n=1):
        """n draws from the cloud by inverse CDF (uniforms from the legacy global RNG)."""
        cdf = self._eng.cumsum(self._weights(), self._norm)
        u = self._eng.to_device(np.random.random((n,)))
        js = self._eng.lw_ancestors(cdf, u)
        return np.ascontiguousarray(self._eng.gather_rows(self._x, js).cpu().numpy().T)
