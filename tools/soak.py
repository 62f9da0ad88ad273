"""Long run: 20000 updates at N = 1e7 (100 passes over the 200-datum schedule, ~3500 resamples each 100 steps):
device memory must stay flat and the step time stable."""
import sys, os, numpy as np, warnings, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
ts = (9 / 8) ** np.arange(200.0)
rs = np.random.RandomState(0)
outs = (rs.random_sample(200) >= np.cos(0.3 * ts / 2) ** 2).astype(int)
upd = qi.SMCUpdater(qi.SimplePrecessionModel(), 10_000_000, qi.UniformDistribution([0, 1]), device_rng=True, seed=0)
free0 = None
for rep in range(100):
    upd.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(200):
        upd.update(int(outs[k]), ts[k:k + 1])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    free, total = torch.cuda.mem_get_info()
    if rep == 2: free0 = free
    if rep % 20 == 0 or rep == 99:
        print("pass %3d: %.4f ms/step, resamples %d, free HBM %.3f GB, torch reserved %.3f GB, mean %.9f" % (
            rep, dt / 200 * 1e3, upd.resample_count, free / 2**30, torch.cuda.memory_reserved() / 2**30, upd.est_mean()[0]))
assert abs(free - free0) < 64 * 2**20, "device memory drifted by %.1f MB" % ((free0 - free) / 2**20)
print("soak ok")
