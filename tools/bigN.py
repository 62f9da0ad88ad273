"""One GPU, N beyond the bucketed sampler's single-pass limit (3.36e7): segmented resample vs the direct sampler."""
import sys, os, numpy as np, warnings, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
n = int(float(os.environ.get("QSMC_N", "1e8")))
for label, limit in (("segmented", 8192 * 4096), ("direct (old behaviour)", 1 << 62)):
    upd = qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0, 1]), device_rng=True, seed=0)
    upd.resampler._segment_limit = limit
    for k in range(4): upd.update(k & 1, np.array([1.5 * (k + 1)]), check_for_resample=False)
    upd.resample(); torch.cuda.synchronize()
    upd.update(0, np.array([7.0]), check_for_resample=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    upd.resample(); torch.cuda.synchronize(); t_rs = time.perf_counter() - t0
    t0 = time.perf_counter()
    for k in range(10): upd.update(k & 1, np.array([9.0 + k]), check_for_resample=False)
    torch.cuda.synchronize(); t_up = (time.perf_counter() - t0) / 10
    print("%-24s N=%.1e  resample %.2f ms   update %.3f ms (%.2e p-u/s)   mean %.6f" % (label, n, t_rs * 1e3, t_up * 1e3, n / t_up, upd.est_mean()[0]))
    del upd; torch.cuda.empty_cache()
