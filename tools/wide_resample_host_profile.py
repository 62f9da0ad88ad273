"""Round 6: where the HOST time of a wide (d = 64) resample goes (cProfile over five resamples, N = 1e6)."""
import os, sys, time, warnings, cProfile, pstats
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-qinfer_amd"))
import torch
import qinfer_amd as qi
warnings.simplefilter("ignore")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000
b = qi.tomography.pauli_basis(3)
m = qi.TomographyModel(b)
np.random.seed(0)
upd = qi.SMCUpdater(m, n, qi.GinibreDistribution(b), device_rng=True, seed=1)
ep = np.zeros((1,), dtype=m.expparams_dtype)
ep["meas"][0, 0] = ep["meas"][0, 5] = np.sqrt(8) / 2
for _ in range(3):
    upd.update(1, ep, check_for_resample=False); upd.resample()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    upd.update(1, ep, check_for_resample=False)
    upd.resample()
torch.cuda.synchronize()
pr.disable()
print("wall per (update + resample): %.0f us" % ((time.perf_counter() - t0) / 5 * 1e6))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
