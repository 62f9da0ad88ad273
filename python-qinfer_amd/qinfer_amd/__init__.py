"""qinfer_amd -- the MI355X-native drop-in for QInfer's SMC hot path.

    import qinfer_amd as qinfer
    updater = qinfer.SMCUpdater(qinfer.SimplePrecessionModel(), 10_000_000,
                                qinfer.UniformDistribution([0, 1]))
    updater.update(outcome, expparams)

Only the path `SMCUpdater.update / batch_update` + `LiuWestResampler` and the plugin surface
around it is provided (see DESIGN.md for scope).  All compute runs in hand-written gfx950 HIP
kernels behind the C ABI of `include/qsmc.h`; importing this package does not need a GPU, using
it does.
"""
from ._exceptions import (ApproximationWarning, NativeLibraryError, ResamplerError,  # noqa: F401
                          ResamplerWarning)
from .abstract_model import FiniteOutcomeModel, Model, Simulatable  # noqa: F401
from .distributions import (Distribution, MultivariateNormalDistribution, ParticleDistribution,  # noqa: F401
                            PostselectedDistribution, ProductDistribution, UniformDistribution)
from .domains import Domain, IntegerDomain  # noqa: F401
from .models import (BinomialModel, DerivedModel, GaussianRandomWalkModel, MLEModel,  # noqa: F401
                     RandomWalkModel, RandomizedBenchmarkingModel,
                     SimpleInversionModel, SimplePrecessionModel, UnknownT2Model)
from .resamplers import LiuWestResampler, Resampler  # noqa: F401
from .smc import SMCUpdater  # noqa: F401
from .simple_est import simple_est_prec, simple_est_rb  # noqa: F401
from .expdesign import EnsembleHeuristic, ExpSparseHeuristic, Heuristic, PGH  # noqa: F401
from .perf_testing import perf_test, perf_test_multiple, timing  # noqa: F401
from . import tomography, utils  # noqa: F401
from .tomography import GinibreDistribution, TomographyModel  # noqa: F401

__version__ = "0.1.0"
