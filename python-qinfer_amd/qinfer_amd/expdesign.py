"""Experiment-choice heuristics: the callers one level above `SMCUpdater.update` that `perf_test`
drives (reference expdesign.py:71-226).  No optimisers -- those stay out of scope.

    ExpSparseHeuristic   t_k = A b^k, k = data seen so far      (the BASELINE schedule, b = 9/8)
    PGH                  particle guess heuristic: two posterior draws -> inversion point and 1 / distance
    EnsembleHeuristic    picks one of several heuristics at random
"""
import abc

import numpy as np

__all__ = ["Heuristic", "EnsembleHeuristic", "ExpSparseHeuristic", "PGH"]


def identity(arg):
    return arg


class Heuristic(metaclass=abc.ABCMeta):
    """Chooses the next experiment from the updater's state, without optimising a risk."""

    def __init__(self, updater):
        self._updater = updater

    @abc.abstractmethod
    def __call__(self, *args):
        raise NotImplementedError("Not yet implemented.")


class EnsembleHeuristic(Heuristic):
    """`ensemble` is a list of (heuristic, probability) pairs; each call delegates to one drawn at random
    (legacy global RNG, `np.random.choice`)."""

    def __init__(self, ensemble):
        self._pr = np.array([pr for _, pr in ensemble])
        self._heuristics = [h for h, _ in ensemble]

    def __call__(self, *args):
        return self._heuristics[np.random.choice(len(self._heuristics), p=self._pr)](*args)


class ExpSparseHeuristic(Heuristic):
    """Exponentially sparse evolution times t_k = scale * base**k with k = len(updater.data_record).
    `t_field=None`: scalar expparams; else the named field of a record, the others from `other_fields`."""

    def __init__(self, updater, scale=1, base=9 / 8, t_field=None, other_fields=None):
        super().__init__(updater)
        self._scale, self._base = scale, base
        self._t_field, self._other_fields = t_field, other_fields

    def __call__(self):
        t = self._scale * (self._base ** len(self._updater.data_record))
        dtype = self._updater.model.expparams_dtype
        if self._t_field is None:
            return np.array([t], dtype=dtype)
        eps = np.empty((1,), dtype=dtype)
        for field, value in (self._other_fields or {}).items():
            eps[field] = value
        eps[self._t_field] = t
        return eps


class PGH(Heuristic):
    """Particle guess heuristic: draw two particles x, x' from the posterior (`updater.sample(2)`, a device
    search); the inversion field gets inv_func(x), the time field t_func(1 / distance(x, x')).  Identical
    draws are retried up to `maxiters` times, then RuntimeError."""

    def __init__(self, updater, inv_field='x_', t_field='t', inv_func=identity, t_func=identity, maxiters=10,
                 other_fields=None):
        super().__init__(updater)
        self._x_, self._t = inv_field, t_field
        self._inv_func, self._t_func = inv_func, t_func
        self._maxiters = maxiters
        self._other_fields = other_fields if other_fields is not None else {}

    def __call__(self):
        model = self._updater.model
        for _ in range(self._maxiters):
            x, xp = self._updater.sample(n=2)[:, np.newaxis, :]
            if model.distance(x, xp) > 0:
                break
        else:
            raise RuntimeError("PGH did not find distinct particles in {} iterations.".format(self._maxiters))
        eps = np.empty((1,), dtype=model.expparams_dtype)
        eps[self._x_] = self._inv_func(x)
        eps[self._t] = self._t_func(1 / model.distance(x, xp))
        for field, value in self._other_fields.items():
            eps[field] = value
        return eps
