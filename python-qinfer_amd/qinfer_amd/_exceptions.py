"""Warning / error classes raised by the SMC path.

Same names and base classes as the reference (`qinfer/_exceptions.py:54-78`) so that user code
doing `except ResamplerError` / `warnings.simplefilter(..., ApproximationWarning)` keeps working.
"""

__all__ = ["ResamplerError", "ResamplerWarning", "ApproximationWarning", "NativeLibraryError"]


class ResamplerError(RuntimeError):
    """A resampler failed in an unrecoverable manner."""

    def __init__(self, msg, cause=None):
        text = msg if cause is None else "{}, caused by exception: {}".format(msg, cause)
        super().__init__(text)
        self._cause = cause


class ResamplerWarning(RuntimeWarning):
    """Something noteworthy (but survivable) happened inside a resampling step."""


class ApproximationWarning(RuntimeWarning):
    """A numerical approximation was violated (negative weights, non-PSD covariance, tiny ESS)."""


class NativeLibraryError(ImportError):
    """libqsmc_hip.so is missing or cannot run: there is deliberately NO CPU fallback."""
