import sys, numpy as np, warnings
sys.path.insert(0,'python-qinfer_amd'); sys.path.insert(0,'oracle')
import qinfer_amd as qi, philox as ph
from qinfer_amd.engine import get_engine
eng=get_engine()
warnings.simplefilter('ignore')
for n in (1<<21, 3000000, 10000000):
    rs=np.random.RandomState(1)
    x=rs.random_sample((n,1)); w=np.cos(x[:,0]/2)**2; w/=w.sum()
    pd=qi.ParticleDistribution(particle_locations=x, particle_weights=w)
    cdf=eng.cumsum(pd._w,1.0).cpu().numpy(); ref=np.cumsum(w)
    print(n,'cdf maxerr',np.abs(cdf-ref).max(), cdf[-1], 'monotone', np.all(np.diff(cdf)>=0))
    u=rs.random_sample(n)
    js=eng.lw_ancestors(eng.to_device(cdf),eng.to_device(u)).cpu().numpy()
    rj=np.minimum(np.searchsorted(ref,u,side='right'),n-1)
    print('  js mismatches',(js!=rj).sum(), 'mean x[js]', x[js,0].mean(), 'expected', np.dot(w,x[:,0]))
    res=qi.LiuWestResampler(a=0.98,device_rng=True,seed=5)
    new=res(qi.SimplePrecessionModel(),pd)
    xn=new.particle_locations[:,0]
    print('  philox resample mean',xn.mean(),'var',xn.var(),'expected var',np.dot(w,x[:,0]**2)-np.dot(w,x[:,0])**2)
    u0,_=ph.uniforms(np.arange(n),5,1,0,0)
    print('  u0 stats',u0.mean(),u0.var())
    ref_new,_=ph.liu_west_philox(w,x,lambda z: z[:,0]>0,0.98,np.sqrt(1-0.98**2),5,1,n)
    print('  vs emulation max diff',np.abs(ref_new[:,0]-xn).max(), 'n diff', (np.abs(ref_new[:,0]-xn)>1e-12).sum())
