import sys, os, numpy as np, warnings, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
m=qi.RandomizedBenchmarkingModel()
prior=qi.PostselectedDistribution(qi.UniformDistribution([[0.8,1],[0,1],[0,1]]), m)
for n in (4096, 12_500_000):
    upd=qi.SMCUpdater(m, n, prior, device_rng=True, resample_thresh=0.0)
    ep=np.empty((1,),dtype=m.expparams_dtype); ep['m']=3
    for _ in range(20): upd.update(0,ep)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(200): upd.update(0,ep)
    torch.cuda.synchronize(); print(n, 'per update us', (time.perf_counter()-t0)/200*1e6)
pr=cProfile.Profile(); pr.enable()
for _ in range(200): upd.update(0,ep)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(8)
