"""Where does the host spend the first resample after a reset?  Wall-clock per call of the pieces of a d = 16 step."""
import os, sys, time, warnings, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qinfer_amd as qi, bench
from qinfer_amd import smc as smc_mod, resamplers as rs_mod, engine as eng_mod, distributions as dist_mod
warnings.simplefilter('ignore')
log = []
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); log.append((name, (time.perf_counter() - t) * 1e6)); return r
    setattr(cls, name, g)
for cls, names in ((smc_mod.SMCUpdater, ["resample", "_step_sync", "_update_step", "_maybe_resample"]),
                   (rs_mod.LiuWestResampler, ["__call__"]), (eng_mod.Engine, ["step", "lw_resample_philox", "sqrtm_psd"]),
                   (dist_mod.ParticleDistribution, ["est_covariance_mtx", "est_mean"])):
    for n in names: wrap(cls, n)
spec = next(s for s in bench.other_config_specs(qi) if s["key"] == "config5_share_tomography")
upd = qi.SMCUpdater(spec["model"], spec["n"], spec["prior"](), device_rng=True, seed=0)
for rep in range(3):
    upd.reset(); torch.cuda.synchronize(); del log[:]
    for k in range(40):
        n0 = len(log); rc = upd.resample_count
        t = time.perf_counter(); upd.update(spec["outs"][k], spec["eps"][k]); dt = (time.perf_counter() - t) * 1e6
        if upd.resample_count != rc or (k and prev_rs):
            print("pass %d datum %2d  update() %.1f us  %s" % (rep, k, dt, "  ".join("%s %.0f" % e for e in log[n0:])))
        prev_rs = upd.resample_count != rc
    torch.cuda.synchronize()
