"""Throughput of the other BASELINE configs' models at scale on one GPU (not bench lines)."""
import sys, os, numpy as np, warnings, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch, qinfer_amd as qi
warnings.simplefilter('ignore')
def run(name, upd, eps, outcomes):
    for k in range(3): upd.update(outcomes[k], eps[k])
    upd.resample(); upd.update(outcomes[2], eps[2]); upd.resample()   # one-time scratch / allocator growth happens here, untimed
    torch.cuda.synchronize(); t0=time.perf_counter(); rc0=upd.resample_count
    for k in range(3, len(eps)): upd.update(outcomes[k], eps[k])
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    n=upd.n_particles; K=len(eps)-3
    print("%-28s N=%.2e  %3d data %2d resamples  %.3f ms/step  %.3e p-u/s  mean[:3]=%s" % (name, n, K, upd.resample_count-rc0, dt/K*1e3, n*K/dt, np.array2string(upd.est_mean()[:3], precision=5)))
rs=np.random.RandomState(0)
# config 3: Binomial(SimplePrecession) n_meas=25, 1e7
m=qi.BinomialModel(qi.SimplePrecessionModel()); K=63
eps=[]; outs=[]
for k in range(K):
    ep=np.empty((1,),dtype=m.expparams_dtype); ep['x']=(9/8)**k; ep['n_meas']=25; eps.append(ep)
    outs.append(int(rs.binomial(25, np.sin(0.3*(9/8)**k/2)**2)))
run("Binomial(Precession) n=25", qi.SMCUpdater(m, 10_000_000, qi.UniformDistribution([0,1]), device_rng=True), eps, outs)
# config 4 (per-GPU share): RB d=3, 1.25e7
m=qi.RandomizedBenchmarkingModel(); K=63
prior=qi.PostselectedDistribution(qi.UniformDistribution([[0.8,1],[0,1],[0,1]]), m)
eps=[]; outs=[]
for k in range(K):
    ep=np.empty((1,),dtype=m.expparams_dtype); ep['m']=1+5*k; eps.append(ep)
    outs.append(int(rs.random_sample() >= 1-(0.3*0.95**(1+5*k)+0.5)))
run("RB (p,A,B)", qi.SMCUpdater(m, 12_500_000, prior, device_rng=True), eps, outs)
# config 5 (per-GPU share): Tomography 2 qubits d=16, 1.25e6, Ginibre prior (host sampled, 1.25e5 tiled x10 for speed)
basis=qi.tomography.pauli_basis(2); m=qi.TomographyModel(basis); K=43
np.random.seed(0); gin=qi.GinibreDistribution(basis); x0=gin.sample(12500); x0=np.tile(x0,(100,1))
class Fixed(qi.Distribution):
    n_rvs=16
    def sample(self,n=1): return x0
true=gin.sample(1)[0]
eps=[]; outs=[]
for k in range(K):
    ep=np.zeros((1,),dtype=m.expparams_dtype); p=rs.randint(1,16); ep['meas'][0,0]=1; ep['meas'][0,p]=1; eps.append(ep)
    outs.append(int(rs.random_sample() < np.clip(true[0]+true[p],0,1)))
run("Tomography 2q (d=16)", qi.SMCUpdater(m, 1_250_000, Fixed(), device_rng=True), eps, outs)
# (f)3 models at scale
K=63; ts=(9/8)**(np.arange(K)*0.5)
outs=[int(rs.random_sample() >= np.cos(0.3*t/2)**2) for t in ts]
m=qi.GaussianRandomWalkModel(qi.SimplePrecessionModel(), fixed_covariance=np.array([1e-8]))
run("GaussianRandomWalk(Precession)", qi.SMCUpdater(m, 10_000_000, qi.UniformDistribution([0,1]), device_rng=True), [ts[k:k+1] for k in range(K)], outs)
m=qi.MLEModel(qi.SimplePrecessionModel(), 2.0)
run("MLEModel(Precession, 2.0)", qi.SMCUpdater(m, 10_000_000, qi.UniformDistribution([0,1]), device_rng=True), [ts[k:k+1] for k in range(K)], outs)
m=qi.UnknownT2Model(); eps=[]
for k in range(K):
    ep=np.empty((1,),dtype=m.expparams_dtype); ep['t']=ts[k]; eps.append(ep)
run("UnknownT2 (d=2)", qi.SMCUpdater(m, 10_000_000, qi.UniformDistribution([[0,1],[0,0.1]]), device_rng=True), eps, outs)
m=qi.BinomialModel(qi.RandomizedBenchmarkingModel()); eps=[]; outs2=[]
prior=qi.PostselectedDistribution(qi.UniformDistribution([[0.8,1],[0,1],[0,1]]), m)
for k in range(K):
    ep=np.empty((1,),dtype=m.expparams_dtype); ep['m']=1+5*k; ep['n_meas']=25; eps.append(ep)
    outs2.append(int(rs.binomial(25, 0.3*0.95**(1+5*k)+0.5)))
run("Binomial(RB) n=25", qi.SMCUpdater(m, 12_500_000, prior, device_rng=True), eps, outs2)
