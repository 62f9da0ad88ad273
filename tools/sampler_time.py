"""Time the resampler's kernels alone on a fixed weighted cloud (N = 1e7 precession by default): the outputs are not
looked at, so ablation builds (QSMC_SAMPLE_ABL) that leave garbage behind can be timed too."""
import os, sys, numpy as np, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
from qinfer_amd.engine import get_engine
warnings.simplefilter('ignore')
n = int(float(os.environ.get("QSMC_N", "1e7")))
eng = get_engine()
upd = qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0, 1]), device_rng=True, seed=0)
for k in range(4):
    upd.update(k & 1, np.array([1.5 * (k + 1)]), check_for_resample=False)
desc = upd.model._native_desc()
mean, cov = upd.est_mean(), upd.est_covariance_mtx()
S, _ = eng.sqrtm_psd(cov, scale=float(np.sqrt(1 - 0.98 ** 2)))
out = eng.empty(1, n)
for rep in range(3):
    eng.lw_resample_philox(desc, True, upd._x, upd._w, upd._norm, 0.98, mean, S, n, 7, rep + 1, 1000, out=out)
torch.cuda.synchronize()
eng.set_profiling(1)
for rep in range(12):
    eng.lw_resample_philox(desc, True, upd._x, upd._w, upd._norm, 0.98, mean, S, n, 7, rep + 10, 1000, out=out)
torch.cuda.synchronize()
ms, tags = eng.profile_read()
eng.set_profiling(0)
print("abl=%s search=%s  sample %.1f us (min %.1f)  counts %.1f us   n_ess/N %.3f" % (
    os.environ.get("QSMC_SAMPLE_ABL"), os.environ.get("QSMC_SAMPLE_BY_SEARCH"), ms[tags == 1].mean() * 1e3,
    ms[tags == 1].min() * 1e3, ms[tags == 6].mean() * 1e3, upd.n_ess / n))
