"""Round 6: where a compiled user model's datum goes, against the native model's (N = 1e7, UnknownT2): updates alone
(no resample test), a forced resample alone, and the per-kernel times of both (HIP events through the profiling ring)."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-qinfer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import qinfer_amd as qi
from qinfer_amd.engine import get_engine
from test_plugin_device import hip_model, t2_data
warnings.simplefilter("ignore")
eng = get_engine()
n = 10_000_000
outcomes, eps = t2_data(40)
for name, model in (("native", qi.UnknownT2Model()), ("hip", hip_model(qi)())):
    upd = qi.SMCUpdater(model, n, qi.UniformDistribution([[0.0, 1.5], [0.0, 0.2]]), device_rng=True, seed=0)
    for k in range(5):
        upd.update(int(outcomes[k]), eps[k:k + 1], check_for_resample=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(5, 25):
        upd.update(int(outcomes[k]), eps[k:k + 1], check_for_resample=False)
    torch.cuda.synchronize(); t_upd = (time.perf_counter() - t0) / 20
    eng.set_profiling(1)
    for k in range(25, 30):
        upd.update(int(outcomes[k]), eps[k:k + 1], check_for_resample=False)
    torch.cuda.synchronize(); ms, tags = eng.profile_read(); eng.set_profiling(0)
    for _ in range(3):                       # (warm: a plugin resample's buffer with spares is allocated at its second call)
        upd.resample(); upd.update(0, eps[0:1], check_for_resample=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        upd.resample(); upd.update(0, eps[3:4], check_for_resample=False)
    torch.cuda.synchronize(); t_rs = (time.perf_counter() - t0) / 5 - t_upd
    print("%-7s update %.1f us/datum (kernel %.1f us), resample %.1f us" % (name, t_upd * 1e6, float(ms[tags != 1].mean()) * 1e3 if len(ms) else -1, t_rs * 1e6))
    # batch_update windows (interval 5, no resample inside the timed part): wall per datum and the window kernel's own time
    upd2 = qi.SMCUpdater(model, n, qi.UniformDistribution([[0.0, 1.5], [0.0, 0.2]]), device_rng=True, seed=0)
    upd2.batch_update(outcomes[:10], eps[:10], resample_interval=5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    upd2.batch_update(outcomes[10:20], eps[10:20], resample_interval=1000)
    torch.cuda.synchronize(); t_win = (time.perf_counter() - t0) / 10
    eng.set_profiling(1)
    upd2.batch_update(outcomes[20:30], eps[20:30], resample_interval=1000)
    torch.cuda.synchronize(); ms, tags = eng.profile_read(); eng.set_profiling(0)
    print("%-7s windows: %.1f us/datum; kernels by tag: %s" % (
        name, t_win * 1e6, {int(t): round(float(ms[tags == t].mean()) * 1e3, 1) for t in np.unique(tags)}))
    del upd, upd2
