"""One-call estimators on top of the GPU SMC path: `simple_est_prec`, `simple_est_rb`
(reference simple_est.py:121-254).

A data table -- a 2-D scalar array with positional columns, a record array / DataFrame with named
columns, or a CSV file -- becomes (outcomes, expparams) for a `BinomialModel`, a fresh `SMCUpdater`
consumes it with `batch_update(..., resample_interval=1)`, and the posterior mean and covariance come
back.  Both models built here have native kernels (`BinomialModel(SimplePrecessionModel)`,
`BinomialModel(RandomizedBenchmarkingModel[interleaved])`), so the whole estimate runs on the device;
with `resample_interval=1` the reference's per-datum n_ess test is kept exactly.
"""
import numpy as np

from .distributions import PostselectedDistribution, UniformDistribution
from .models import BinomialModel, RandomizedBenchmarkingModel, SimplePrecessionModel
from .smc import SMCUpdater

try:                                   # optional, like the reference: DataFrames are accepted if pandas exists
    import pandas as pd
except Exception:  # noqa: BLE001
    pd = None

__all__ = ["simple_est_prec", "simple_est_rb", "data_to_params", "load_data_or_txt"]


def _is_scalar_dtype(dt):
    """True for a plain scalar type spec (the old `np.issctype`): not a field list, not a string name."""
    if not isinstance(dt, (type, np.dtype)):
        return False
    try:
        t = np.dtype(dt).type
    except TypeError:
        return False
    return t is not np.object_ and issubclass(t, np.generic)


def data_to_params(data, expparams_dtype, col_outcomes=(0, 'counts'), cols_expparams=None):
    """Split a data table into the outcomes column and an expparams array (simple_est.py:69-106).

    Every column is named by a pair (index, field name): the index is used for homogeneous 2-D arrays,
    the name for record arrays.  `cols_expparams` maps expparams fields to such pairs (or is one pair
    if `expparams_dtype` is a scalar type)."""
    homogeneous = _is_scalar_dtype(data.dtype) and not data.dtype.fields
    pick = (lambda col: data[..., col[0]]) if homogeneous else (lambda col: data[col[1]])
    outcomes = pick(col_outcomes).astype(int)
    expparams = np.empty(outcomes.shape, dtype=expparams_dtype)
    if _is_scalar_dtype(expparams_dtype):
        expparams[:] = pick(cols_expparams)
    else:
        for field, col in cols_expparams.items():
            expparams[field] = pick(col)
    return outcomes, expparams


def load_data_or_txt(data, dtype):
    """ndarray -> itself; DataFrame -> records; filename / file object -> np.loadtxt(..., delimiter=',')."""
    if isinstance(data, np.ndarray):
        return data
    if pd is not None and isinstance(data, pd.DataFrame):
        return data.to_records(index=False)
    if hasattr(data, 'read') or isinstance(data, str):
        return np.loadtxt(data, dtype=dtype, delimiter=',')
    raise TypeError("Expected a filename, an array or a file-like object.")


def do_update(model, n_particles, prior, outcomes, expparams, return_all, resampler=None, **updater_kwargs):
    updater = SMCUpdater(model, n_particles, prior, resampler=resampler, **updater_kwargs)
    updater.batch_update(outcomes, expparams, resample_interval=1)
    mean = updater.est_mean()
    cov = updater.est_covariance_mtx()
    if model.n_modelparams == 1:
        mean, cov = mean[0], cov[0, 0]
    if not return_all:
        return mean, cov
    return mean, cov, {'updater': updater}


def simple_est_prec(data, freq_min=0.0, freq_max=1.0, n_particles=6000, return_all=False, **updater_kwargs):
    """Frequency of a cos^2 precession from rows (counts, t, n_shots): posterior mean and variance
    (and `{'updater': ...}` with `return_all`).  Extra keyword arguments go to `SMCUpdater`
    (e.g. `device_rng=True, seed=...`)."""
    model = BinomialModel(SimplePrecessionModel(freq_min))
    prior = UniformDistribution([0, freq_max])
    data = load_data_or_txt(data, [('counts', 'uint'), ('t', float), ('n_shots', 'uint')])
    outcomes, expparams = data_to_params(data, model.expparams_dtype,
                                         cols_expparams={'x': (1, 't'), 'n_meas': (2, 'n_shots')})
    return do_update(model, n_particles, prior, outcomes, expparams, return_all, **updater_kwargs)


def simple_est_rb(data, interleaved=False, p_min=0.0, p_max=1.0, n_particles=8000, return_all=False,
                  **updater_kwargs):
    """Randomized-benchmarking parameters (p, A, B) -- or (p_tilde, p_ref, A, B) with `interleaved` --
    from rows (counts, m, n_shots[, reference]): posterior mean vector and covariance matrix."""
    model = BinomialModel(RandomizedBenchmarkingModel(interleaved=interleaved))
    box = [[p_min, p_max], [0, 1], [0, 1]] if not interleaved else [[p_min, p_max], [p_min, p_max], [0, 1], [0, 1]]
    prior = PostselectedDistribution(UniformDistribution(box), model)
    data = load_data_or_txt(data, [('counts', 'uint'), ('m', 'uint'), ('n_shots', 'uint')] +
                            ([('reference', 'uint')] if interleaved else []))
    cols = {'m': (1, 'm'), 'n_meas': (2, 'n_shots')}
    if interleaved:
        cols['reference'] = (3, 'reference')
    outcomes, expparams = data_to_params(data, model.expparams_dtype, cols_expparams=cols)
    return do_update(model, n_particles, prior, outcomes, expparams, return_all, **updater_kwargs)
