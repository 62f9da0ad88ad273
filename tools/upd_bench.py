import sys, os, numpy as np, warnings, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
from qinfer_amd import _native
from qinfer_amd.engine import get_engine
eng=get_engine(); warnings.simplefilter('ignore')
n=int(float(os.environ.get('N','1e7')))
upd=qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0,1]), device_rng=True, seed=1)
desc=upd.model._native_desc(); w2=eng.empty(n); ep=_native.make_expparam(t=37.0)
eng.set_profiling(True)
ms=[]
for i in range(60):
    eng.update_fused(desc, upd._x, upd._w, w2, 1.0, ep, 0)
    ms.append(eng.last_update_kernel_ms())
ms=np.array(ms[10:])
print(os.environ.get('QSMC_NT','0'), os.environ.get('QSMC_GRID_CAP','2048'), 'kernel us median %.2f min %.2f -> %.0f GB/s' % (np.median(ms)*1e3, ms.min()*1e3, 24*n/np.median(ms)/1e6))
