import sys, os, numpy as np, warnings, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
n=10_000_000
ts=(9/8)**np.arange(200.0); rs=np.random.RandomState(0)
outcomes=(rs.random_sample(200) >= np.cos(0.3*ts/2)**2).astype(int)
for interval in (1,5,8):
  for fastp in (True, False):
    upd=qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0,1]), device_rng=True, seed=0)
    upd._batch_fast_path=fastp
    upd.batch_update(outcomes[:20], ts[:20], resample_interval=interval); upd.reset(); upd._resample_count=0
    torch.cuda.synchronize(); t0=time.perf_counter()
    upd.batch_update(outcomes, ts, resample_interval=interval)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("batch_update interval=%d fused=%s: %.3f ms/datum  %.3e p-u/s  resamples=%d mean=%.6f" % (interval, fastp, dt/200*1e3, n*200/dt, upd.resample_count, upd.est_mean()[0]))
