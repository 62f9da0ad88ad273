"""Model plugin surface of the SMC engine.

Same class names, method names, argument meaning and array shapes as the reference's
`qinfer/abstract_model.py` (Simulatable :66, Model :397, FiniteOutcomeModel :544), restated so a
user's `Model` subclass written against QInfer drops in unchanged:

    likelihood(outcomes, modelparams, expparams) -> float64 L[n_outcomes, n_models, n_experiments]
    are_models_valid(modelparams)                -> bool  [n_models]
    simulate_experiment(modelparams, expparams, repeat=1)
    update_timestep(modelparams, expparams)      -> [n_models, n_modelparams, n_experiments]
    canonicalize(modelparams)                    -> [n_models, n_modelparams]

Models that additionally implement the *native* hooks

    _native_desc()                -> _native.ModelDesc         (which HIP kernel family)
    _native_expparams(expparams)  -> list of _native.ExpParam  (one per experiment)

are served entirely by the HIP kernels; `NativeModelMixin` then also routes the NumPy-contract
`likelihood` / `are_models_valid` through the same kernels, so there is a single source of truth
for the model arithmetic.  A model without the hooks still works with `SMCUpdater` through the
plugin slow path (its own `likelihood` runs on the host, the weight update stays on the GPU).

Device-side plugin hooks (optional; a user model written against torch tensors stays in HBM -- the updater then
makes no host copy of the cloud at all).  `x_dev` is the cloud as the updater holds it: a float64 torch tensor of
shape (n_modelparams, n_particles) on the GPU, structure-of-arrays (row m = parameter m of every particle); treat
it as read-only.

    likelihood_device(outcomes, x_dev, expparams)   -> float64 device tensor L[n_outcomes, n_experiments, n_particles]
                                                       (the contract of `likelihood`, abstract_model.py:444-468, with
                                                       the particle axis last)
    are_models_valid_device(x_dev)                  -> bool / uint8 device tensor [n_particles]
    update_timestep_device(x_dev, expparams)        -> float64 device tensor (n_modelparams, n_particles): the cloud
                                                       after one experiment's time step
    canonicalize_device(x_dev)                      -> float64 device tensor (n_modelparams, n_particles)

Each is used in place of its NumPy namesake when present (and the model is not served by native kernels); a model
may define any subset.  `Model.count_likelihood_calls` keeps `call_count` meaningful from `likelihood_device`.

A user model at the speed of a native one: `likelihood_hip` (a class or instance attribute, a string of HIP device
source).  The updater compiles it with hiprtc INTO the fused update kernel (qsmc_user_kernel_build,
csrc/kernels/user_jit.hpp): one pass of 16 + 8 d bytes per particle per datum, like the library's own models.

    likelihood_hip = r'''
    __device__ double likelihood(const double *x, const double *ep, long long outcome) {   // Pr(outcome | x; ep)
        const double t = ep[0], e = exp(-t * x[1]), c = cos(0.5 * x[0] * t);
        const double pr0 = e * c * c + 0.5 * (1.0 - e);
        return outcome == 0 ? pr0 : 1.0 - pr0;
    }
    #define QSMC_USER_HAS_VALID 1                                                          // optional: are_models_valid
    __device__ bool valid(const double *x) { return x[0] >= 0.0 && x[1] >= 0.0; }
    '''

`x[0 .. QSMC_D)` is one particle, `ep[0 .. QSMC_NEP)` the experiment record's fields as doubles in dtype order (vector
fields flattened; a plain-dtype experiment is one double).  The NumPy methods stay the contract (simulate_experiment and
the host-side callers use them); the source must state the same function.
"""
import abc

import numpy as np

from .domains import IntegerDomain

__all__ = ["Simulatable", "Model", "FiniteOutcomeModel", "NativeModelMixin", "native_ok"]


def safe_shape(arr, idx=0, default=1):
    shape = np.shape(arr)
    return shape[idx] if idx < len(shape) else default


class Simulatable(metaclass=abc.ABCMeta):
    """Something that produces data given model parameters and experiment parameters."""

    def __init__(self):
        self._sim_count = 0
        self._Q = np.ones((self.n_modelparams,))

    # -- abstract ------------------------------------------------------------------------
    @property
    @abc.abstractmethod
    def n_modelparams(self):
        """Number of real model parameters."""

    @property
    @abc.abstractmethod
    def expparams_dtype(self):
        """NumPy dtype (scalar name or record spec) of one experiment."""

    @abc.abstractmethod
    def n_outcomes(self, expparams):
        """Number of outcomes per experiment (a scalar if constant)."""

    @abc.abstractmethod
    def domain(self, expparams):
        """List of outcome domains, one per experiment (or one Domain for expparams=None)."""

    @abc.abstractmethod
    def are_models_valid(self, modelparams):
        """bool[n_models]: which parameter vectors are admissible."""

    @abc.abstractmethod
    def simulate_experiment(self, modelparams, expparams, repeat=1):
        self._sim_count += modelparams.shape[0] * expparams.shape[0] * repeat
        assert self.are_expparam_dtypes_consistent(expparams)

    # -- concrete ------------------------------------------------------------------------
    @property
    def is_n_outcomes_constant(self):
        return True

    @property
    def model_chain(self):
        return ()

    @property
    def base_model(self):
        return self

    @property
    def underlying_model(self):
        chain = self.model_chain
        return chain[-1] if chain else None

    @property
    def sim_count(self):
        return self._sim_count

    @property
    def Q(self):
        return self._Q

    @property
    def modelparam_names(self):
        return ["x_{{{}}}".format(i) for i in range(self.n_modelparams)]

    def are_expparam_dtypes_consistent(self, expparams):
        if self.is_n_outcomes_constant:
            return True
        if expparams.size == 0:
            return True
        doms = self.domain(expparams)
        return all(dm.dtype == doms[0].dtype for dm in doms[1:])

    def clear_cache(self):
        """Nothing cached by default."""

    def experiment_cost(self, expparams):
        return np.ones(expparams.shape)

    def distance(self, a, b):
        return np.sum(np.abs(self.Q * (a - b)), axis=1)

    def update_timestep(self, modelparams, expparams):
        """Static parameters: x(t_{k+1}) = x(t_k), shape (n_models, n_modelparams, n_experiments)."""
        return np.repeat(np.asarray(modelparams)[:, :, np.newaxis], expparams.shape[0], axis=2)

    def canonicalize(self, modelparams):
        return modelparams


class Model(Simulatable):
    """A Simulatable that can also evaluate likelihoods."""

    def __init__(self, allow_identical_outcomes=False, outcome_warning_threshold=0.99):
        super().__init__()
        self._call_count = 0
        self._allow_identical_outcomes = allow_identical_outcomes
        self._outcome_warning_threshold = outcome_warning_threshold

    @property
    def call_count(self):
        return self._call_count

    @property
    def allow_identical_outcomes(self):
        return self._allow_identical_outcomes

    @allow_identical_outcomes.setter
    def allow_identical_outcomes(self, value):
        self._allow_identical_outcomes = value

    @property
    def outcome_warning_threshold(self):
        return self._outcome_warning_threshold

    @outcome_warning_threshold.setter
    def outcome_warning_threshold(self, value):
        self._outcome_warning_threshold = value

    @abc.abstractmethod
    def likelihood(self, outcomes, modelparams, expparams):
        """L[i, j, k] = Pr(outcomes[i] | modelparams[j]; expparams[k]).  Subclasses call this
        base implementation to keep `call_count` (abstract_model.py:466-468) meaningful."""
        self._call_count += safe_shape(outcomes) * safe_shape(modelparams) * safe_shape(expparams)

    def is_model_valid(self, modelparams):
        return bool(self.are_models_valid(np.asarray(modelparams)[np.newaxis, :])[0])

    def count_likelihood_calls(self, n_outcomes, n_models, n_experiments):
        """What the base `likelihood` adds to `call_count` (abstract_model.py:466-468), for a `likelihood_device`
        that never sees host arrays."""
        self._call_count += int(n_outcomes) * int(n_models) * int(n_experiments)


class FiniteOutcomeModel(Model):
    """Models whose outcomes are integers 0 .. n_outcomes-1."""

    def __init__(self, allow_identical_outcomes=False, outcome_warning_threshold=0.99,
                 n_outcomes_cutoff=None):
        super().__init__(allow_identical_outcomes=allow_identical_outcomes,
                         outcome_warning_threshold=outcome_warning_threshold)
        self._n_outcomes_cutoff = n_outcomes_cutoff
        if self.is_n_outcomes_constant:
            self._domain = IntegerDomain(min=0, max=self.n_outcomes(None) - 1)

    @property
    def n_outcomes_cutoff(self):
        return self._n_outcomes_cutoff

    @n_outcomes_cutoff.setter
    def n_outcomes_cutoff(self, value):
        self._n_outcomes_cutoff = value

    def domain(self, expparams):
        if self.is_n_outcomes_constant:
            return self._domain if expparams is None else [self._domain for _ in expparams]
        return [IntegerDomain(min=0, max=int(n) - 1) for n in self.n_outcomes(expparams)]

    def simulate_experiment(self, modelparams, expparams, repeat=1):
        """Inverse-CDF sampling of outcomes with the global legacy RNG, one uniform per
        (repeat, model, experiment) exactly as abstract_model.py:632-658 consumes them."""
        super().simulate_experiment(modelparams, expparams, repeat)
        n_m, n_e = modelparams.shape[0], expparams.shape[0]
        if self.is_n_outcomes_constant:
            values = self.domain(None).values
            cdf = np.cumsum(self.likelihood(values, modelparams, expparams), axis=0)
            rnd = np.random.random((repeat, 1, n_m, n_e))
            outcomes = values[np.argmax(cdf > rnd, axis=1)]
        else:
            dtype = self.domain(expparams[0, np.newaxis])[0].dtype
            outcomes = np.empty((repeat, n_m, n_e), dtype=dtype)
            for k in range(n_e):
                one = expparams[k:k + 1]
                values = self.domain(one)[0].values
                cdf = np.cumsum(self.likelihood(values, modelparams, one), axis=0)[..., 0]
                rnd = np.random.random((repeat, 1, n_m))
                outcomes[:, :, k] = values[np.argmax(cdf > rnd, axis=1)]
        if repeat == 1 and n_e == 1 and n_m == 1:
            return outcomes[0, 0, 0]
        return outcomes

    @staticmethod
    def pr0_to_likelihood_array(outcomes, pr0):
        """(n_models, n_experiments) Pr(0) -> L[n_outcomes, n_models, n_experiments]."""
        pr0 = np.asarray(pr0)[np.newaxis, ...]
        pr1 = 1 - pr0
        outcomes = np.atleast_1d(np.asarray(outcomes))
        return np.concatenate([pr0 if outcomes[i] == 0 else pr1 for i in range(outcomes.shape[0])])


# The methods whose arithmetic the HIP kernels of a native model stand for.  A subclass written outside this
# package that overrides any of them means something else by the model than the kernels compute: it is served by
# the plugin path (its own methods run on the host), like any other user `Model`.
_KERNEL_BACKED = ("likelihood", "are_models_valid", "update_timestep", "canonicalize", "n_outcomes", "domain")


_OWN_MODULES = frozenset(__name__.rsplit(".", 1)[0] + "." + m for m in ("abstract_model", "models", "tomography"))
_CLASS_OK = {}


def _class_is_native(cls):
    """Every kernel-backed method of `cls` is DEFINED by a class of this library (looked up along the MRO, so
    properties, partials and builtins -- which carry no __module__ of their own -- are judged by the class that
    holds them).  Cached per class."""
    ok = _CLASS_OK.get(cls)
    if ok is None:
        ok = True
        for name in _KERNEL_BACKED:
            owner = next((k for k in cls.__mro__ if name in vars(k)), None)
            if owner is not None and owner.__module__ not in _OWN_MODULES:
                ok = False
                break
        _CLASS_OK[cls] = ok
    return ok


def native_ok(model):
    """True if `model` is served by the HIP kernels: it declares native hooks AND every kernel-backed method of
    its class is this library's own implementation (the reference dispatches on the overriding method,
    abstract_model.py:444-528; a user override must win here too)."""
    if model is None or not getattr(model, "_native", False):
        return False
    return _class_is_native(type(model))


class NativeModelMixin:
    """Routes the NumPy-contract methods of a model through its HIP kernels.

    Subclasses provide `_native_desc()` and `_native_expparams(expparams)`.
    """

    _native = True

    def _engine(self):
        from .engine import get_engine
        return get_engine()

    def _native_expparams(self, expparams):  # pragma: no cover - abstract hook
        raise NotImplementedError

    def _native_desc(self):  # pragma: no cover - abstract hook
        raise NotImplementedError

    def _native_likelihood(self, outcomes, modelparams, expparams):
        eng = self._engine()
        mp = np.asarray(modelparams, dtype=np.float64)
        if mp.ndim == 1:
            mp = mp[:, np.newaxis]
        outcomes = np.atleast_1d(np.asarray(outcomes)).astype(np.int64).ravel()
        x = eng.locs_to_soa(mp)
        L = eng.likelihood(self._native_desc(), x, self._native_expparams(expparams), outcomes)
        return np.ascontiguousarray(L.cpu().numpy().transpose(0, 2, 1))      # (n_o, N, n_e)

    def _native_are_models_valid(self, modelparams):
        eng = self._engine()
        mp = np.asarray(modelparams, dtype=np.float64)
        x = eng.locs_to_soa(mp)
        return eng.are_models_valid(self._native_desc(), x).cpu().numpy().astype(bool)
