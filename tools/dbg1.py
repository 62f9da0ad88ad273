import sys, numpy as np
sys.path.insert(0,'python-qinfer_amd'); sys.path.insert(0,'oracle')
import qinfer_amd as qi
from qinfer_amd.engine import get_engine
eng=get_engine()
for n in (300000, 600000, 1000000, 3000000, 10000000):
    rs=np.random.RandomState(1)
    x=rs.random_sample((n,1)); w=np.cos(x[:,0]/2)**2; w/=w.sum()
    pd=qi.ParticleDistribution(particle_locations=x, particle_weights=w)
    m=pd.est_mean()[0]; ref=np.dot(w,x[:,0])
    s0,s1,s2=eng.moments(pd._x,pd._w,1.0)
    print(n, m, ref, m-ref, s0, s2[0,0]-np.dot(w,x[:,0]**2))
