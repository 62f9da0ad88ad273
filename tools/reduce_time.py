"""Updates only (no resample) at N = 1e7: for timing k_reduce_partials variants under rocprofv3 --stats."""
import sys, os, warnings, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
upd = qi.SMCUpdater(qi.SimplePrecessionModel(), 10_000_000, qi.UniformDistribution([0, 1]), device_rng=True, resample_thresh=0.0)
t = np.array([0.01])
for _ in range(20): upd.update(0, t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): upd.update(0, t)
torch.cuda.synchronize(); print('us/step', (time.perf_counter() - t0) / 300 * 1e6)
