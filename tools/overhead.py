import sys, os, numpy as np, warnings, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
upd=qi.SMCUpdater(qi.SimplePrecessionModel(), 4096, qi.UniformDistribution([0,1]), device_rng=True, resample_thresh=0.0)
t=np.array([0.01]); 
for _ in range(100): upd.update(0,t)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(2000): upd.update(0,t)
torch.cuda.synchronize(); print('per update overhead us', (time.perf_counter()-t0)/2000*1e6)
pr=cProfile.Profile(); pr.enable()
for _ in range(2000): upd.update(0,t)
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
