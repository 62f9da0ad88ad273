"""`perf_test` / `perf_test_multiple`: the trial loop that defines the benchmark's metric
(reference perf_testing.py:182-384).

One trial = draw a true model, then n_exp times: heuristic -> simulated datum -> timed
`SMCUpdater.update` -> loss of the posterior mean.  The record array has the reference's fields
(loss, resample_count, elapsed_time, outcome, true, est, experiment fields).  Trials are independent, so
several GPUs run them as replicas with no communication: `perf_test_multiple(..., comm=group)` gives rank r
the trials r, r + G, ... and gathers the records once at the end.
"""
import contextlib
import time
from functools import partial

import numpy as np
import numpy.ma as ma

from .smc import SMCUpdater

__all__ = ["timing", "Timer", "numpy_err_policy", "perf_test", "perf_test_multiple", "actual_dtype",
           "PERFORMANCE_DTYPE", "apply_serial"]


class Timer:
    """Wall-clock time since construction; frozen by stop()."""

    def __init__(self):
        self._tic, self._toc = time.time(), None

    def stop(self):
        self._toc = time.time()

    @property
    def delta_t(self):
        return (self._toc if self._toc is not None else time.time()) - self._tic

    def __repr__(self):
        return "<qinfer_amd.Timer at 0x{:x}, {:.6g} s elapsed>".format(id(self), self.delta_t)


@contextlib.contextmanager
def timing():
    """`with timing() as t: ...` then `t.delta_t` seconds.  `SMCUpdater.update` returns with its sums read
    back, so the block's wall time includes the kernels it launched for that datum (a resample triggered
    by the datum is asynchronous and is charged to the next datum's synchronisation)."""
    t = Timer()
    yield t
    t.stop()


@contextlib.contextmanager
def numpy_err_policy(**kwargs):
    old = np.seterr(**kwargs)
    yield
    np.seterr(**old)


PERFORMANCE_DTYPE = [('loss', float), ('resample_count', int), ('elapsed_time', float), ('outcome', int)]


def actual_dtype(model, true_model=None):
    true_model = model if true_model is None else true_model
    model_dtype = [('true', float, true_model.n_modelparams), ('est', float, model.n_modelparams)]
    if isinstance(model.expparams_dtype, str):
        return PERFORMANCE_DTYPE + model_dtype + [('experiment', model.expparams_dtype)], True
    return PERFORMANCE_DTYPE + model_dtype + model.expparams_dtype, False


def _promote_dims_left(a, ndim):
    a = np.asarray(a)
    return a.reshape((1,) * max(0, ndim - a.ndim) + a.shape)


def _shorten_right(*args):
    """Trim every array to the shortest trailing shape (models with different parameter counts align right)."""
    arrs = [np.asarray(a) for a in args]
    m = min(a.shape[-1] for a in arrs)
    return tuple(a[..., -m:] for a in arrs)


class _TrialColumns:
    """Per-datum results of one trial, column by column in plain arrays; the record array of the reference's layout
    (perf_testing.py:182-218 dtype) is assembled from them once, at the end of the trial."""

    def __init__(self, model, true_model, n_exp):
        self.dtype, self.scalar_experiment = actual_dtype(model, true_model)
        self.experiment_fields = [] if self.scalar_experiment else [f[0] for f in model.expparams_dtype]
        self.elapsed = np.zeros(n_exp)
        self.loss = np.zeros(n_exp)
        self.resample_count = np.zeros(n_exp, dtype=int)
        self.outcome = np.zeros(n_exp, dtype=int)
        self.true = np.zeros((n_exp, true_model.n_modelparams))
        self.est = np.zeros((n_exp, model.n_modelparams))
        self.experiments = [None] * n_exp

    def record(self):
        rec = np.zeros((len(self.elapsed),), dtype=self.dtype)
        rec['elapsed_time'], rec['loss'] = self.elapsed, self.loss
        rec['resample_count'], rec['outcome'] = self.resample_count, self.outcome
        rec['true'], rec['est'] = self.true, self.est
        if self.scalar_experiment:
            rec['experiment'] = np.concatenate([np.ravel(e) for e in self.experiments])
        else:
            for name in self.experiment_fields:
                rec[name] = np.concatenate([np.atleast_1d(e[name]) for e in self.experiments]).reshape(rec[name].shape)
        return rec


def _data_stream(updater, heuristic, true_model, true_mps, n_exp):
    """The experiment loop of a trial as a generator: design -> simulated datum -> TIMED update -> the true model's own
    time step.  Yields (experiment, datum, seconds in `update`, true parameters before / after the datum)."""
    for _ in range(n_exp):
        expparams = heuristic()
        datum = true_model.simulate_experiment(true_mps, expparams)
        with timing() as t:
            updater.update(datum, expparams)
        before = true_mps
        true_mps = true_model.update_timestep(_promote_dims_left(true_mps, 2), expparams)[:, :, 0]
        yield expparams, datum, t.delta_t, before, true_mps


def perf_test(model, n_particles, prior, n_exp, heuristic_class, true_model=None, true_prior=None, true_mps=None,
              extra_updater_args=None):
    """One trial; returns a record array of length n_exp (see module docstring)."""
    true_model = model if true_model is None else true_model
    true_mps = (prior if true_prior is None else true_prior).sample() if true_mps is None else true_mps
    updater = SMCUpdater(model, n_particles, prior, **({} if extra_updater_args is None else extra_updater_args))
    cols = _TrialColumns(model, true_model, n_exp)
    q_tail = model.Q[-min(model.n_modelparams, true_model.n_modelparams):]
    stream = _data_stream(updater, heuristic_class(updater), true_model, true_mps, n_exp)
    for k, (expparams, datum, seconds, true_before, true_after) in enumerate(stream):
        cols.true[k] = true_before                   # (per datum: the true model may itself drift)
        cols.experiments[k], cols.outcome[k], cols.elapsed[k] = expparams, datum, seconds
        cols.est[k] = updater.est_mean()
        est, tru = _shorten_right(cols.est[k], true_after)
        cols.loss[k] = np.dot(np.subtract(est, tru) ** 2, q_tail)
        cols.resample_count[k] = updater.resample_count
    return cols.record()


class apply_serial:
    """Runs fn in this process when .get() is called: the calling convention of a parallel engine's `apply`."""

    def __init__(self, fn, *args, **kwargs):
        self._fn, self._args, self._kwargs = fn, args, kwargs
        self._done, self._value = False, None

    def get(self):
        if not self._done:
            self._value, self._done = self._fn(*self._args, **self._kwargs), True
        return self._value


def _finished_trials(indices, apply, trial_fn):
    """Dispatch every trial through `apply` first (a parallel engine starts them all), then hand back
    (trial index, record or the exception it raised) as they are collected, in dispatch order."""
    handles = [(idx, apply(trial_fn)) for idx in indices]
    for idx, handle in handles:
        try:
            yield idx, handle.get()
        except Exception as exc:  # noqa: BLE001
            yield idx, exc


def perf_test_multiple(n_trials, model, n_particles, prior, n_exp, heuristic_class, true_model=None,
                       true_prior=None, true_mps=None, apply=apply_serial, allow_failures=False,
                       extra_updater_args=None, progressbar=None, comm=None):
    """n_trials independent trials -> record array (n_trials, n_exp), masked rows for failed trials with
    `allow_failures`.  `comm` (a ParticleShardGroup or any object with rank / world_size / dist): trials are
    dealt to the ranks round-robin and the records all-gathered -- replicas, no data-path collective."""
    trial_fn = partial(perf_test, model, n_particles, prior, n_exp, heuristic_class, true_model, true_prior,
                       true_mps=true_mps, extra_updater_args=extra_updater_args)
    dtype, _ = actual_dtype(model, true_model)
    performance = (ma.zeros if allow_failures else np.zeros)((n_trials, n_exp), dtype=dtype)
    rank, world = (0, 1) if comm is None else (comm.rank, comm.world_size)
    mine = list(range(rank, n_trials, world))
    prog = progressbar() if progressbar is not None else None
    try:
        if prog is not None:
            prog.start(len(mine))
        with numpy_err_policy(divide='raise'):
            for done, (idx, outcome) in enumerate(_finished_trials(mine, apply, trial_fn)):
                if isinstance(outcome, Exception):
                    if not allow_failures:
                        raise outcome
                    performance.mask[idx, :] = True
                    continue
                performance[idx, :] = outcome
                if prog is not None:
                    prog.update(done)
    finally:
        if prog is not None:
            prog.finished()
    if comm is not None and world > 1:
        parts = [None] * world
        comm.dist.all_gather_object(parts, (mine, performance[mine]), group=getattr(comm, "group", None))
        for idxs, rows in parts:
            if len(idxs):
                performance[idxs] = rows
    return performance
