"""Host-side cost of a tomography update at small N (cProfile)."""
import sys, os, numpy as np, warnings, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
from qinfer_amd import tomography as tomo
warnings.simplefilter('ignore')
basis = tomo.pauli_basis(2)
m = tomo.TomographyModel(basis)
prior = tomo.GinibreDistribution(basis)
upd = qi.SMCUpdater(m, 4096, prior, device_rng=True, resample_thresh=0.0)
ep = np.zeros(1, dtype=m.expparams_dtype); ep['meas'][0, 0] = 0.5; ep['meas'][0, 3] = 0.5
for _ in range(100): upd.update(0, ep)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): upd.update(0, ep)
torch.cuda.synchronize(); print('per update us', (time.perf_counter() - t0) / 2000 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): upd.update(0, ep)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(14)
