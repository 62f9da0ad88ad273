import sys, os, numpy as np, warnings, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
upd=qi.SMCUpdater(qi.SimplePrecessionModel(), 4096, qi.UniformDistribution([0,1]), device_rng=True, resample_thresh=1.0)
t=np.array([0.3]);
for _ in range(100): upd.update(0,t)
torch.cuda.synchronize(); t0=time.perf_counter(); c0=upd.resample_count
for _ in range(2000): upd.update(0,t)
torch.cuda.synchronize(); print('per update+resample us', (time.perf_counter()-t0)/2000*1e6, 'resamples', upd.resample_count-c0)
pr=cProfile.Profile(); pr.enable()
for _ in range(2000): upd.update(0,t)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(22)
