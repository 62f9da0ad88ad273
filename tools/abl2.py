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
