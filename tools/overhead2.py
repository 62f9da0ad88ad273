import sys, os, numpy as np, warnings, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
upd=qi.SMCUpdater(qi.SimplePrecessionModel(), 100000, qi.UniformDistribution([0.2,0.8]), device_rng=True)
t=np.array([1.0])
upd.update(0,t,check_for_resample=False)
for _ in range(20): upd.update(0,t,check_for_resample=False); upd.resample()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(300): upd.update(0,t,check_for_resample=False); upd.resample()
torch.cuda.synchronize(); print('update+resample us', (time.perf_counter()-t0)/300*1e6)
pr=cProfile.Profile(); pr.enable()
for _ in range(300): upd.update(0,t,check_for_resample=False); upd.resample()
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(32)
