import sys, os, numpy as np, warnings, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
rs=np.random.RandomState(0)
basis=qi.tomography.pauli_basis(2); m=qi.TomographyModel(basis); K=43
np.random.seed(0); gin=qi.GinibreDistribution(basis); x0=gin.sample(12500); x0=np.tile(x0,(100,1))
class Fixed(qi.Distribution):
    n_rvs=16
    def sample(self,n=1): return x0
true=gin.sample(1)[0]
upd=qi.SMCUpdater(m, 1_250_000, Fixed(), device_rng=True)
for k in range(K):
    ep=np.zeros((1,),dtype=m.expparams_dtype); p=rs.randint(1,16); ep['meas'][0,0]=1; ep['meas'][0,p]=1
    upd.update(int(rs.random_sample() < np.clip(true[0]+true[p],0,1)), ep)
torch.cuda.synchronize(); print(upd.resample_count)
