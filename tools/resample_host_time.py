"""Where the host time of a resample step goes (N = 1e7 precession, per-datum update): wall time of each host segment
between the update's synchronisation and the launch of the sampling kernel, accumulated over the resamples of the schedule."""
import sys, os, time, warnings, collections, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qinfer_amd as qi
from qinfer_amd import resamplers, smc, distributions, engine
warnings.simplefilter('ignore')
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(obj, name, label=None):
    f = getattr(obj, name); label = label or name
    def g(*a, **k):
        t0 = time.perf_counter_ns()
        try: return f(*a, **k)
        finally:
            e = acc[label]; e[0] += 1; e[1] += (time.perf_counter_ns() - t0) * 1e-3
    setattr(obj, name, g)
cfg = next((a.split('=')[1] for a in sys.argv[1:] if a.startswith('--config=')), None)
if cfg:                                            # one of bench.py's other configs (config4_share_rb, config5_share_tomography ...)
    import bench
    spec = [s_ for s_ in bench.other_config_specs(qi) if s_['key'] == cfg][0]
    upd = qi.SMCUpdater(spec['model'], spec['n'], spec['prior'](), device_rng=True, seed=0)
    outcomes, eps = spec['outs'], spec['eps']
else:
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    upd = qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0, 1]), device_rng=True, seed=0)
    rng = np.random.default_rng(0)
    ts = (9 / 8) ** np.arange(200); outcomes = (rng.random(200) < np.sin(0.3 * ts / 2) ** 2).astype(int)
    eps = [ts[k:k + 1] for k in range(200)]
K = len(eps)
for k in range(min(40, K)): upd.update(int(outcomes[k]), eps[k])
upd.reset(); upd._resample_count = 0
wrap(smc.SMCUpdater, 'resample'); wrap(smc.SMCUpdater, '_maybe_resample')
wrap(resamplers.LiuWestResampler, '__call__', 'resampler.__call__'); wrap(resamplers.LiuWestResampler, '_prepare_device')
wrap(distributions.ParticleDistribution, 'est_mean'); wrap(distributions.ParticleDistribution, 'est_covariance_mtx')
wrap(engine.Engine, 'sqrtm_psd'); wrap(engine.Engine, 'lw_resample_philox'); wrap(engine.Engine, 'lw_resample_prepare')
wrap(engine.Engine, 'update_fused'); wrap(smc.SMCUpdater, '_canonicalize_device')
wrap(distributions.ParticleDistribution, '_from_device')
wrap(smc.SMCUpdater, 'update')
class LibProxy:
    """times the C entry points themselves (the ctypes call), apart from the Python around them"""
    def __init__(self, lib): self.__dict__['_lib'] = lib
    def __getattr__(self, name):
        f = getattr(self._lib, name)
        def g(*a):
            t0 = time.perf_counter_ns()
            try: return f(*a)
            finally:
                e = acc['C:' + name]; e[0] += 1; e[1] += (time.perf_counter_ns() - t0) * 1e-3
        self.__dict__[name] = g
        return g
upd._eng.lib = LibProxy(upd._eng.lib)
wrap(np.linalg, 'eigvals', 'np.linalg.eigvals'); wrap(np.linalg, 'eigh', 'np.linalg.eigh'); wrap(warnings, 'warn', 'warnings.warn')
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(K): upd.update(int(outcomes[k]), eps[k])
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print('ms/step (instrumented)', wall / K * 1e3, 'resamples', upd.resample_count)
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%-28s n=%4d  mean %7.2f us  total %8.1f us' % (k, c, t / c, t))
