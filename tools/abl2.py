import sys, os, numpy as np, warnings, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch
from qinfer_amd import _native
if os.environ.get('QSMC_ABL_LIB'):
    _native._LIB_PATH = os.environ['QSMC_ABL_LIB']
import qinfer_amd as qi
from qinfer_amd.engine import get_engine
eng=get_engine(); warnings.simplefilter('ignore')
n=int(os.environ.get("QSMC_N","10000000"))
upd=qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0.2,0.8]), device_rng=True, seed=1)
for t in (3.0, 5.0, 9.0): upd.update(0,np.array([t]),check_for_resample=False)
mean=upd.est_mean(); cov=upd.est_covariance_mtx(); S,_=eng.sqrtm_psd(cov,0.2)
desc=upd.model._native_desc()
for _ in range(3): eng.lw_resample_philox(desc,True,upd._x,upd._w,upd._norm,0.98,mean,S,n,1,1,1000)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): eng.lw_resample_philox(desc,True,upd._x,upd._w,upd._norm,0.98,mean,S,n,1,1,1000)
torch.cuda.synchronize(); print(os.environ.get('QSMC_ABL_LIB','default'), (time.perf_counter()-t)/10*1e6,'us per resample call')
try:
    import ctypes as C
    lib = eng.lib
    fn = lib.qsmc_dbg_phase
    fn.argtypes = [C.POINTER(C.c_double)]; fn.restype = C.c_int
    out = (C.c_double * 8)()
    fn(out)                                     # reset
    for _ in range(1): eng.lw_resample_philox(desc,True,upd._x,upd._w,upd._norm,0.98,mean,S,n,1,1,1000)
    torch.cuda.synchronize(); fn(out)
    nb = out[2]
    print('scan_block: load+local scan %.2f us, barrier wait %.2f us, clamp+max+store+guide %.2f us' % (out[3]/nb/100, out[4]/nb/100, out[5]/nb/100))
    print('phase: blocks', nb, ' setup %.2f us  pairloop %.2f us (per workgroup, thread 0, 100 MHz clock)' % (out[0] / nb / 100, out[1] / nb / 100))
except AttributeError:
    pass
