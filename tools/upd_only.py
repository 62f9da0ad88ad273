import sys, os, numpy as np, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch
from qinfer_amd import _native
if os.environ.get('QSMC_ABL_LIB'): _native._LIB_PATH = os.environ['QSMC_ABL_LIB']
import qinfer_amd as qi
warnings.simplefilter('ignore')
upd=qi.SMCUpdater(qi.SimplePrecessionModel(), 10_000_000, qi.UniformDistribution([0,1]), device_rng=True, resample_thresh=0.0)
for k in range(12): upd.update(k&1, np.array([1.125**(3*k)]))
torch.cuda.synchronize()
