"""How many outputs of each RB resample (config 4 share) ask for a global redraw."""
import sys, os, warnings, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 12_500_000
m = qi.RandomizedBenchmarkingModel()
prior = qi.PostselectedDistribution(qi.UniformDistribution([[0.8, 1], [0, 1], [0, 1]]), m)
upd = qi.SMCUpdater(m, n, prior, device_rng=True, seed=0)
rng = np.random.default_rng(0)
rc = 0
for k in range(60):
    mm = 1 + 5 * k
    p0 = 1 - (0.3 * 0.95 ** mm + 0.5)
    upd.update(int(rng.random() >= p0), np.array([(mm,)], dtype=m.expparams_dtype))
    if upd.resample_count != rc:
        rc = upd.resample_count
        upd.update(0, np.array([(mm,)], dtype=m.expparams_dtype), check_for_resample=False)   # next sync publishes the count
        print('datum', k, 'resample', rc, 'redraws', upd._eng.last_resample_redraws(), 'of', n)
