"""Outcome domains.  Only the integer domain is needed on the SMC path (two-outcome and
binomial models); mirrors the interface of `qinfer/domains.py` IntegerDomain (min/max/values/
n_members/dtype/in_domain) that `FiniteOutcomeModel.simulate_experiment` relies on."""
import numpy as np

__all__ = ["Domain", "IntegerDomain"]


class Domain:
    """Marker base class for outcome domains."""


class IntegerDomain(Domain):
    def __init__(self, min=0, max=np.inf):
        self._min = int(min) if np.isfinite(min) else min
        self._max = int(max) if np.isfinite(max) else max

    @property
    def min(self):
        return self._min

    @property
    def max(self):
        return self._max

    @property
    def is_continuous(self):
        return False

    @property
    def is_discrete(self):
        return True

    @property
    def is_finite(self):
        return bool(np.isfinite(self._min) and np.isfinite(self._max))

    @property
    def dtype(self):
        return np.dtype(int)

    @property
    def n_members(self):
        return int(self._max - self._min + 1) if self.is_finite else np.inf

    @property
    def example_point(self):
        return np.array([self._min if np.isfinite(self._min) else 0], dtype=self.dtype)

    @property
    def values(self):
        if not self.is_finite:
            raise ValueError("an unbounded integer domain has no finite list of values")
        return np.arange(self._min, self._max + 1, dtype=self.dtype)

    def in_domain(self, points):
        pts = np.asarray(points)
        return bool(np.all(pts >= self._min) and np.all(pts <= self._max)
                    and np.all(np.mod(pts, 1) == 0))

    def __repr__(self):
        return "IntegerDomain(min={}, max={})".format(self._min, self._max)
