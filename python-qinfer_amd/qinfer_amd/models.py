"""Concrete models on the SMC hot path, each backed by a HIP likelihood kernel.

    SimpleInversionModel / SimplePrecessionModel   reference test_models.py:64-213
    DerivedModel / BinomialModel                   reference derived_models.py:82-144, 222-360
    RandomizedBenchmarkingModel                    reference rb.py:81-195

Class names, constructor arguments, `expparams_dtype` record layouts, `modelparam_names`, validity
rules and the `likelihood` tensor shape are the reference's.  The arithmetic lives in
`csrc/qsmc_device.h` (one `Model<KIND>` specialisation per class); the methods here only translate
NumPy `expparams` into the C-ABI's `qsmc_expparam_t`.
"""
import numpy as np

from . import _native
from .abstract_model import FiniteOutcomeModel, Model, NativeModelMixin, native_ok
from .domains import IntegerDomain

__all__ = ["SimpleInversionModel", "SimplePrecessionModel", "UnknownT2Model", "DerivedModel", "BinomialModel",
           "MLEModel", "RandomWalkModel", "GaussianRandomWalkModel", "RandomizedBenchmarkingModel"]


def _field(expparams, name):
    return np.atleast_1d(expparams[name]).ravel()


class SimpleInversionModel(NativeModelMixin, FiniteOutcomeModel):
    r"""Qubit precessing under H = omega sigma_z / 2, inverted by w_ before measurement:
    Pr(0 | omega; t, w_) = cos^2(t (omega - w_) / 2).  Valid iff omega > min_freq."""

    def __init__(self, min_freq=0):
        super().__init__()
        self._min_freq = min_freq

    @property
    def n_modelparams(self):
        return 1

    @property
    def modelparam_names(self):
        return [r'\omega']

    @property
    def expparams_dtype(self):
        return [('t', 'float'), ('w_', 'float')]

    @property
    def is_n_outcomes_constant(self):
        return True

    def n_outcomes(self, expparams):
        return 2

    # native hooks
    def _native_desc(self):
        return _native.ModelDesc(_native.MODEL_PRECESSION, 1, float(self._min_freq), 0, 0)

    def _native_expparams(self, expparams):
        expparams = np.atleast_1d(expparams)
        ts, ws = _field(expparams, 't'), _field(expparams, 'w_')
        return [_native.make_expparam(t=t, w_=w) for t, w in zip(ts, ws)]

    # NumPy contract, served by the kernels
    def are_models_valid(self, modelparams):
        return self._native_are_models_valid(modelparams)

    def likelihood(self, outcomes, modelparams, expparams):
        super().likelihood(outcomes, modelparams, expparams)
        return self._native_likelihood(outcomes, modelparams, expparams)


class SimplePrecessionModel(SimpleInversionModel):
    r"""SimpleInversionModel with w_ = 0 and a scalar experiment parameter t."""

    @property
    def expparams_dtype(self):
        return 'float'

    def _native_fill_expparam(self, ep, expparams):
        """Per-datum path of SMCUpdater.update: write the one experiment into the updater's own qsmc_expparam_t
        (no allocation).  False: not the plain shape -- take `_native_expparams`."""
        if type(expparams) is np.ndarray and expparams.shape == (1,) and expparams.dtype.names is None:
            ep.t = expparams[0]
            return True
        return False

    def _native_expparams(self, expparams):
        if type(expparams) is np.ndarray and expparams.shape == (1,) and expparams.dtype.names is None:
            return [_native.make_expparam(t=expparams[0], w_=0.0)]       # the per-datum path of update()
        expparams = np.atleast_1d(expparams)
        ts = expparams['t'] if expparams.dtype.names else expparams
        return [_native.make_expparam(t=t, w_=0.0) for t in np.atleast_1d(ts).astype(np.float64).ravel()]


class UnknownT2Model(NativeModelMixin, FiniteOutcomeModel):
    r"""Qubit prepared in |+>, precessing under H = omega sigma_z / 2 with an unknown dephasing rate:
    Pr(0 | omega, 1/T2; t) = e^{-t/T2} cos^2(omega t / 2) + (1 - e^{-t/T2}) / 2
    (reference test_models.py:222-259).  Valid iff both parameters are >= 0."""

    @property
    def n_modelparams(self):
        return 2

    @property
    def modelparam_names(self):
        return [r'\omega', r'T_2^{-1}']

    @property
    def expparams_dtype(self):
        return [('t', 'float')]

    @property
    def is_n_outcomes_constant(self):
        return True

    def n_outcomes(self, expparams):
        return 2

    def _native_desc(self):
        return _native.ModelDesc(_native.MODEL_UNKNOWN_T2, 2, 0.0, 0, 0)

    def _native_expparams(self, expparams):
        expparams = np.atleast_1d(expparams)
        return [_native.make_expparam(t=t) for t in _field(expparams, 't')]

    def are_models_valid(self, modelparams):
        return self._native_are_models_valid(modelparams)

    def likelihood(self, outcomes, modelparams, expparams):
        super().likelihood(outcomes, modelparams, expparams)
        return self._native_likelihood(outcomes, modelparams, expparams)


class DerivedModel(Model):
    """Base for models that decorate another model: passes everything through by default."""

    _underlying_model = None

    def __init__(self, underlying_model):
        self._underlying_model = underlying_model
        super().__init__()

    @property
    def underlying_model(self):
        return self._underlying_model

    @property
    def base_model(self):
        return self._underlying_model.base_model

    @property
    def model_chain(self):
        return self._underlying_model.model_chain + (self._underlying_model,)

    @property
    def n_modelparams(self):
        return self._underlying_model.n_modelparams

    @property
    def expparams_dtype(self):
        return self._underlying_model.expparams_dtype

    @property
    def modelparam_names(self):
        return self._underlying_model.modelparam_names

    @property
    def Q(self):
        return self._underlying_model.Q

    def clear_cache(self):
        self._underlying_model.clear_cache()

    def n_outcomes(self, expparams):
        return self._underlying_model.n_outcomes(expparams)

    def are_models_valid(self, modelparams):
        return self._underlying_model.are_models_valid(modelparams)

    def domain(self, expparams):
        return self._underlying_model.domain(expparams)

    def are_expparam_dtypes_consistent(self, expparams):
        return self._underlying_model.are_expparam_dtypes_consistent(expparams)

    def update_timestep(self, modelparams, expparams):
        return self._underlying_model.update_timestep(modelparams, expparams)

    def canonicalize(self, modelparams):
        return self._underlying_model.canonicalize(modelparams)

    def simulate_experiment(self, modelparams, expparams, repeat=1):
        return self._underlying_model.simulate_experiment(modelparams, expparams, repeat)


class BinomialModel(NativeModelMixin, DerivedModel):
    """n_meas i.i.d. shots of a two-outcome model: L[k] = Binom(n_meas, pr1).pmf(k), where pr1 is
    the underlying model's likelihood of outcome 1.  Adds the `n_meas` experiment field (and names
    a scalar underlying experiment parameter `x`).

    Native (fully on the GPU) when the decorated model is a SimplePrecessionModel (the BASELINE
    config-3 model, and what `simple_est_prec` builds) or a RandomizedBenchmarkingModel, plain or
    interleaved (what `simple_est_rb` builds).  Other two-outcome models go through the plugin slow path.
    """

    def __init__(self, underlying_model):
        super().__init__(underlying_model)
        if not (underlying_model.is_n_outcomes_constant and underlying_model.n_outcomes(None) == 2):
            raise ValueError("Decorated model must be a two-outcome model.")
        if isinstance(underlying_model.expparams_dtype, str):
            self._expparams_scalar = True
            self._expparams_dtype = [('x', underlying_model.expparams_dtype), ('n_meas', 'uint')]
        else:
            self._expparams_scalar = False
            self._expparams_dtype = underlying_model.expparams_dtype + [('n_meas', 'uint')]
        # kernels exist for Binomial(SimplePrecession) and Binomial(RB[interleaved]); the same rule as every other
        # decorator decides whether the underlying model is the library's own (abstract_model.native_ok)
        self._native = (native_ok(underlying_model)
                        and isinstance(underlying_model, (SimplePrecessionModel, RandomizedBenchmarkingModel)))
        self._um_is_rb = isinstance(underlying_model, RandomizedBenchmarkingModel)

    @property
    def decorated_model(self):
        return self.underlying_model

    @property
    def expparams_dtype(self):
        return self._expparams_dtype

    @property
    def is_n_outcomes_constant(self):
        return False

    def n_outcomes(self, expparams):
        return expparams['n_meas'] + 1

    def domain(self, expparams):
        return [IntegerDomain(min=0, max=int(n) - 1) for n in np.atleast_1d(self.n_outcomes(expparams))]

    def are_expparam_dtypes_consistent(self, expparams):
        return True

    def _underlying_expparams(self, expparams):
        return expparams['x'] if self._expparams_scalar else expparams

    # native hooks
    def _native_desc(self):
        um = self.underlying_model
        if self._um_is_rb:
            kind = _native.MODEL_BINOMIAL_RB_INTERLEAVED if um._il else _native.MODEL_BINOMIAL_RB
            return _native.ModelDesc(kind, um.n_modelparams, 0.0, 0, 0)
        return _native.ModelDesc(_native.MODEL_BINOMIAL_PRECESSION, 1, float(um._min_freq), 0, 0)

    def _native_fill_expparam(self, ep, expparams):
        if type(expparams) is np.ndarray and expparams.shape == (1,) and not self._um_is_rb:
            e = expparams[0]
            ep.t, ep.n_meas = e['x'], int(e['n_meas'])
            return True
        return False

    def _native_expparams(self, expparams):
        um = self.underlying_model
        if type(expparams) is np.ndarray and expparams.shape == (1,) and not self._um_is_rb:
            e = expparams[0]                                           # the per-datum path of update()
            return [_native.make_expparam(t=e['x'], w_=0.0, n_meas=e['n_meas'])]
        expparams = np.atleast_1d(expparams)
        ns = _field(expparams, 'n_meas')
        if self._um_is_rb:
            ms = _field(expparams, 'm')
            refs = _field(expparams, 'reference') if um._il else np.zeros(ms.shape, dtype=bool)
            return [_native.make_expparam(m=m, reference=int(bool(r)), n_meas=n) for m, r, n in zip(ms, refs, ns)]
        return [_native.make_expparam(t=t, w_=0.0, n_meas=n) for t, n in zip(_field(expparams, 'x'), ns)]

    def likelihood(self, outcomes, modelparams, expparams):
        Model.likelihood(self, outcomes, modelparams, expparams)
        if self._native:
            return self._native_likelihood(outcomes, modelparams, expparams)
        # plugin slow path for an arbitrary decorated model (host arithmetic, like any user Model)
        from scipy.stats import binom
        pr1 = self.underlying_model.likelihood(np.array([1], dtype='uint'), modelparams,
                                               self._underlying_expparams(expparams))
        outcomes = np.atleast_1d(outcomes)
        return np.concatenate([binom(expparams['n_meas'][np.newaxis, :], pr1).pmf(outcomes[i])
                               for i in range(outcomes.shape[0])])

    def are_models_valid(self, modelparams):
        if self._native:
            return self._native_are_models_valid(modelparams)
        return self.underlying_model.are_models_valid(modelparams)

    def simulate_experiment(self, modelparams, expparams, repeat=1):
        """Binomial draws from pr1 (legacy global RNG, like derived_models.py:331-355)."""
        pr1 = self.underlying_model.likelihood(np.array([1], dtype='uint'), modelparams,
                                               self._underlying_expparams(expparams))[0]
        n = expparams['n_meas'].astype('int')[np.newaxis, :]
        os_ = np.stack([np.random.binomial(np.broadcast_to(n, pr1.shape), pr1) for _ in range(repeat)])
        return os_[0, 0, 0] if os_.size == 1 else os_

    def update_timestep(self, modelparams, expparams):
        return self.underlying_model.update_timestep(modelparams, self._underlying_expparams(expparams))


class MLEModel(NativeModelMixin, DerivedModel):
    r"""Approximate maximum-likelihood estimation by amplifying the Bayes update: every likelihood
    call of the decorated model is raised to `likelihood_power` (reference derived_models.py:673-691).

    Native when the decorated model is: the power is applied inside the same kernels (`model_lik`),
    so updating, batch updating and resampling cost what they cost for the decorated model."""

    def __init__(self, underlying_model, likelihood_power):
        super().__init__(underlying_model)
        self._pow = likelihood_power
        self._native = native_ok(underlying_model)
        # a decorated random-walk model still walks (derived_models.py:177-178 forwards update_timestep): hand its
        # device step through, so that update() does not have to choose between the kernel path and the walk
        step = getattr(underlying_model, "_native_timestep", None)
        self._native_timestep = step if (self._native and step is not None) else None

    @property
    def is_n_outcomes_constant(self):
        return self.underlying_model.is_n_outcomes_constant

    def _native_desc(self):
        desc = self.underlying_model._native_desc()
        prev = desc.likelihood_power if desc.likelihood_power != 0.0 else 1.0      # nested MLEModels multiply
        return _native.ModelDesc(desc.kind, desc.d, desc.min_freq, desc.postselect_all_valid, desc.reserved,
                                 float(prev * self._pow))

    def _native_expparams(self, expparams):
        return self.underlying_model._native_expparams(expparams)

    def likelihood(self, outcomes, modelparams, expparams):
        if self._native:
            Model.likelihood(self, outcomes, modelparams, expparams)
            return self._native_likelihood(outcomes, modelparams, expparams)
        return self.underlying_model.likelihood(outcomes, modelparams, expparams) ** self._pow

    def are_models_valid(self, modelparams):
        return self.underlying_model.are_models_valid(modelparams)

    def simulate_experiment(self, modelparams, expparams, repeat=1):
        return self.underlying_model.simulate_experiment(modelparams, expparams, repeat)


class _WalkingModel(NativeModelMixin, DerivedModel):
    """Shared plumbing of the random-walk decorators: likelihood, validity and simulation are the decorated
    model's (on its first n parameters); the walk happens in `update_timestep` / `_native_timestep`."""

    _walk_epoch = 0

    @property
    def is_n_outcomes_constant(self):
        return self.underlying_model.is_n_outcomes_constant

    def _native_desc(self):
        return self.underlying_model._native_desc()

    def _native_expparams(self, expparams):
        return self.underlying_model._native_expparams(expparams)

    def _walk_seed(self, updater):
        self._walk_epoch += 1
        rank = 0 if updater._comm is None else updater._comm.rank
        return updater._seed + 0x9E3779B97F4A7C15 * (rank + 1), self._walk_epoch


class RandomWalkModel(_WalkingModel):
    r"""After every datum each particle takes a step drawn from `step_distribution`
    (reference derived_models.py:693-741).  The steps are sampled by the distribution (host, any law) and
    added to the device-resident cloud in place -- the locations never travel to the host."""

    def __init__(self, underlying_model, step_distribution):
        self._step_dist = step_distribution
        super().__init__(underlying_model)
        if self.underlying_model.n_modelparams != self._step_dist.n_rvs:
            raise TypeError("Step distribution does not match model dimension.")
        self._native = native_ok(underlying_model)

    def likelihood(self, outcomes, modelparams, expparams):
        Model.likelihood(self, outcomes, modelparams, expparams)
        return self.underlying_model.likelihood(outcomes, modelparams, expparams)

    def simulate_experiment(self, modelparams, expparams, repeat=1):
        return self.underlying_model.simulate_experiment(modelparams, expparams, repeat)

    def update_timestep(self, modelparams, expparams):
        # the step is independent of the experiment; one step vector per (particle, experiment)
        n, n_e = modelparams.shape[0], expparams.shape[0]
        steps = self._step_dist.sample(n=n * n_e).reshape((n, n_e, self.n_modelparams)).transpose((0, 2, 1))
        return modelparams[:, :, np.newaxis] + steps

    def _native_timestep(self, updater, expparams):
        eng = updater._eng
        steps = np.asarray(self._step_dist.sample(n=updater.n_particles), dtype=np.float64)     # (n, d)
        eng.random_walk(updater._x, np.ones(self.n_modelparams), z=eng.locs_to_soa(steps))


class _StepLaw:
    """How the walking parameters of a GaussianRandomWalkModel step: one object per covariance variant.  `draw(mp, n_e)`
    returns unit-scale steps of shape (n, n_rw, n_e) from the legacy global RNG; `extra_names` are the model parameters
    the law appends (none when the covariance is given); `admissible(mp)` is its share of are_models_valid;
    `unit_covariance(mp)` the covariance of one step."""
    extra_names = ()
    device_scale = None                 # per-parameter step sigma if the law is a fixed diagonal (the device kernel's case)

    def bind(self, first_extra_column):
        self.cols = first_extra_column + np.arange(len(self.extra_names))

    def admissible(self, mp):
        return None


class _KnownDiagonal(_StepLaw):
    def __init__(self, variances, n_rw):
        v = np.asarray(variances, dtype=np.float64)
        if v.ndim != 1:
            raise ValueError('Diagonal covariance requested, but fixed_covariance has {} dimensions.'.format(v.ndim))
        if v.size != n_rw:
            raise ValueError('fixed_covariance dimension, {}, inconsistent with number of parameters, {}'
                             .format(v.size, n_rw))
        self.device_scale = np.sqrt(v)

    def draw(self, mp, n_e):
        # draw order pinned by fixture G10 (the reference draws one (n_eps, n_mps, n_rw) block: derived_models.py:929)
        z = np.random.normal(size=(n_e, mp.shape[0], self.device_scale.size))
        return np.moveaxis(z * self.device_scale, 0, 2)

    def unit_covariance(self, mp):
        return np.diag(self.device_scale ** 2)


class _KnownDense(_StepLaw):
    def __init__(self, cov, n_rw):
        cov = np.asarray(cov, dtype=np.float64)
        if cov.ndim != 2:
            raise ValueError('Dense covariance requested, but fixed_covariance has {} dimensions.'.format(cov.ndim))
        if cov.shape != (n_rw, n_rw):
            raise ValueError('fixed_covariance expected to be square with width {}'.format(n_rw))
        self.factor = np.linalg.cholesky(cov)           # cov = factor factor^T

    def draw(self, mp, n_e):
        # stream order of the reference's dense-fixed branch (derived_models.py:932-935): ONE block of unit normals with a
        # row per walking parameter and a column per (particle, experiment) pair, particle-major -- coloured from the left
        # by the factor -- so that a seeded legacy-RNG run consumes np.random exactly as QInfer does (fixture g10 dense case)
        n, n_rw = mp.shape[0], self.factor.shape[0]
        white = np.random.normal(size=(n_rw, n * n_e))
        coloured = self.factor @ white                                   # (n_rw, n * n_e)
        return np.moveaxis(coloured.reshape(n_rw, n, n_e), 0, 1)       # (n, n_rw, n_e)

    def unit_covariance(self, mp):
        return self.factor @ self.factor.T


class _LearnedDiagonal(_StepLaw):
    """One sigma per walking parameter rides in the cloud; sigma >= 0 is part of validity."""

    def __init__(self, names):
        self.extra_names = tuple(r"\sigma_{{{}}}".format(nm) for nm in names)

    def draw(self, mp, n_e):
        sig = mp[:, self.cols]
        z = np.random.normal(size=(n_e,) + sig.shape)
        return np.moveaxis(z * sig, 0, 2)

    def admissible(self, mp):
        return (mp[..., self.cols] >= 0).all(axis=-1)

    def unit_covariance(self, mp):
        return np.diag((mp[:, self.cols] ** 2).mean(axis=0))


class _LearnedDense(_StepLaw):
    """The lower triangle of a Cholesky-like factor per particle rides in the cloud (row-major over the triangle)."""

    def __init__(self, names):
        k = len(names)
        self.rows, self.cols_in_factor = np.tril_indices(k)
        self.k = k
        self.extra_names = tuple(
            r"\sigma_{{{}}}".format(names[i]) if i == j else r"\sigma_{{{},{}}}".format(names[j], names[i])
            for i, j in zip(self.rows, self.cols_in_factor))

    def _factors(self, mp):
        f = np.zeros((mp.shape[0], self.k, self.k))
        f[:, self.rows, self.cols_in_factor] = mp[:, self.cols]
        return f

    def draw(self, mp, n_e):
        z = np.random.normal(size=(mp.shape[0], self.k, n_e))
        return self._factors(mp) @ z

    def unit_covariance(self, mp):
        f = self._factors(mp)
        return (f @ np.swapaxes(f, 1, 2)).mean(axis=0)


class GaussianRandomWalkModel(_WalkingModel):
    r"""After every datum the parameters selected by `random_walk_idxs` take a zero-mean Gaussian step (the contract of
    reference derived_models.py:743-963, written from SURVEY 8(f)3's description).  The step covariance is either given
    (`fixed_covariance`: a vector of variances if `diagonal`, else a full matrix) or learned, in which case the entries
    of its square root are appended to the model parameters and each particle walks with its own belief.
    `scale_mult` (a function of expparams, or the name of an expparams field) scales the step of a given experiment;
    `model_transformation = (f, f_inv)` applies the walk in transformed coordinates.

    With a given diagonal covariance, no transformation and a decorated model that has kernels -- the case SURVEY
    8(f)3 names -- the whole step runs on the device (`qsmc_random_walk`): Philox normals with `device_rng`, else
    the reference's own draw uploaded once (fixture G10 pins that stream).  The other variants are outside 8(f)3: they
    run on the host (plugin slow path) with the right law but no fixture pins their draw order."""

    def __init__(self, underlying_model, random_walk_idxs='all', fixed_covariance=None, diagonal=True,
                 scale_mult=None, model_transformation=None):
        n_u = underlying_model.n_modelparams
        pick = slice(None) if (isinstance(random_walk_idxs, str) and random_walk_idxs == 'all') else random_walk_idxs
        self._walkers = np.atleast_1d(np.arange(n_u)[pick])              # columns of the decorated model that walk
        if self._walkers.size == 0:
            raise IndexError('At least one model parameter must take a random walk.')
        names = [underlying_model.modelparam_names[i] for i in self._walkers]
        if fixed_covariance is not None:
            law = (_KnownDiagonal if diagonal else _KnownDense)(fixed_covariance, len(names))
        else:
            law = (_LearnedDiagonal if diagonal else _LearnedDense)(names)
        law.bind(n_u)
        self._law = law
        super().__init__(underlying_model)
        if scale_mult is None:
            self._step_multiplier = lambda expparams: 1.0
        elif isinstance(scale_mult, str):
            self._step_multiplier = lambda expparams, field=scale_mult: expparams[field]
        else:
            self._step_multiplier = scale_mult
        self._coords = model_transformation            # (to_walk_coords, from_walk_coords) or None
        self._native = bool(native_ok(underlying_model) and law.device_scale is not None and self._coords is None)
        if not self._native:
            self._native_timestep = None            # (instance attribute shadows the method: plugin slow path)

    # ------------------------------------------------------------------ surface
    @property
    def modelparam_names(self):
        return self.underlying_model.modelparam_names + list(self._law.extra_names)

    @property
    def n_modelparams(self):
        return self.underlying_model.n_modelparams + len(self._law.extra_names)

    @property
    def is_n_outcomes_constant(self):
        return False

    def _u(self, modelparams):
        return modelparams[..., :self.underlying_model.n_modelparams]

    def are_models_valid(self, modelparams):
        ok = self.underlying_model.are_models_valid(self._u(modelparams))
        extra = self._law.admissible(modelparams)
        return ok if extra is None else np.logical_and(ok, extra)

    def likelihood(self, outcomes, modelparams, expparams):
        Model.likelihood(self, outcomes, modelparams, expparams)
        return self.underlying_model.likelihood(outcomes, self._u(modelparams), expparams)

    def simulate_experiment(self, modelparams, expparams, repeat=1):
        return self.underlying_model.simulate_experiment(self._u(modelparams), expparams, repeat)

    def est_update_covariance(self, modelparams):
        """Covariance of one unit step (its particle average when it is being learned)."""
        return self._law.unit_covariance(np.asarray(modelparams))

    def update_timestep(self, modelparams, expparams):
        """Host arithmetic, every variant (legacy global RNG): (n, n_modelparams, n_expparams)."""
        modelparams = np.asarray(modelparams, dtype=np.float64)
        n_e = expparams.shape[0]
        steps = self._law.draw(modelparams, n_e) * self._step_multiplier(expparams)        # (n, n_rw, n_e)
        n_u = self.underlying_model.n_modelparams
        out = np.empty(modelparams.shape + (n_e,))
        for e in range(n_e):
            cur = modelparams.copy()
            if self._coords is not None:
                cur[:, :n_u] = self._coords[0](cur[:, :n_u])
            cur[:, self._walkers] += steps[:, :, e]
            if self._coords is not None:
                cur[:, :n_u] = self._coords[1](cur[:, :n_u])
            out[:, :, e] = cur
        return out

    def _native_timestep(self, updater, expparams):
        """Given diagonal covariance: the step of one datum, in place on the device."""
        eng = updater._eng
        mult = float(np.ravel(self._step_multiplier(np.atleast_1d(expparams)))[0])
        scale = np.zeros(self.n_modelparams)
        scale[self._walkers] = self._law.device_scale * mult
        if updater._device_rng:
            seed, epoch = self._walk_seed(updater)
            eng.random_walk(updater._x, scale, z=None, seed=seed, epoch=epoch)
        else:
            # parity mode: the draw `update_timestep` would make for one experiment, uploaded once
            z = np.random.normal(size=(1, updater.n_particles, self._walkers.size))[0]
            eng.random_walk(updater._x, scale, z=eng.locs_to_soa(z))


class RandomizedBenchmarkingModel(NativeModelMixin, FiniteOutcomeModel):
    r"""Zeroth-order (interleaved) randomized benchmarking: Pr(0) = 1 - (A p^m + B).

    Model parameters (p, A, B), or (p_tilde, p_ref, A, B) when `interleaved=True`, in which case
    the experiment field `reference` selects p_ref (True) or p_tilde * p_ref (False)."""

    def __init__(self, interleaved=False, order=0):
        self._il = bool(interleaved)
        if order != 0:
            raise NotImplementedError("Only zeroth-order is currently implemented.")
        super().__init__()

    @property
    def n_modelparams(self):
        return 4 if self._il else 3

    @property
    def modelparam_names(self):
        return [r'\tilde{p}', 'p', 'A', 'B'] if self._il else ['p', 'A', 'B']

    @property
    def is_n_outcomes_constant(self):
        return True

    @property
    def expparams_dtype(self):
        return [('m', 'uint')] + ([('reference', bool)] if self._il else [])

    def n_outcomes(self, expparams):
        return 2

    def _native_desc(self):
        kind = _native.MODEL_RB_INTERLEAVED if self._il else _native.MODEL_RB
        return _native.ModelDesc(kind, self.n_modelparams, 0.0, 0, 0)

    def _native_fill_expparam(self, ep, expparams):
        if type(expparams) is np.ndarray and expparams.shape == (1,) and not self._il:
            ep.m = int(expparams[0]['m'])
            return True
        return False

    def _native_expparams(self, expparams):
        if type(expparams) is np.ndarray and expparams.shape == (1,) and not self._il:
            return [_native.make_expparam(m=expparams[0]['m'], reference=0)]     # the per-datum path of update()
        expparams = np.atleast_1d(expparams)
        ms = _field(expparams, 'm')
        refs = _field(expparams, 'reference') if self._il else np.zeros(ms.shape, dtype=bool)
        return [_native.make_expparam(m=m, reference=int(bool(r))) for m, r in zip(ms, refs)]

    def are_models_valid(self, modelparams):
        return self._native_are_models_valid(modelparams)

    def likelihood(self, outcomes, modelparams, expparams):
        super().likelihood(outcomes, modelparams, expparams)
        return self._native_likelihood(outcomes, modelparams, expparams)
