"""Update kernel alone (event-timed) at a few cloud sizes: in the Infinity Cache (1e7) and beyond it (3e7, 1e8)."""
import os, sys, numpy as np, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
from qinfer_amd.engine import get_engine
warnings.simplefilter('ignore')
eng = get_engine()
for n in (10_000_000, 30_000_000, 100_000_000):
    upd = qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0, 1]), device_rng=True, seed=0)
    for k in range(3): upd.update(k & 1, np.array([1.5 * (k + 1)]), check_for_resample=False)
    torch.cuda.synchronize(); eng.set_profiling(1)
    for k in range(16): upd.update(k & 1, np.array([2.0 + 0.37 * k]), check_for_resample=False)
    torch.cuda.synchronize(); ms, tags = eng.profile_read(); eng.set_profiling(0)
    us = ms[tags == 0].mean() * 1e3
    print("%s N=%.0e  update kernel %.1f us  %.0f GB/s  frac %.3f" % (os.environ.get("QSMC_LIB_PATH", "in-tree")[-10:], n, us, 24 * n / us / 1e3, 24 * n / us / 1e3 / 8000))
    del upd; torch.cuda.empty_cache()
