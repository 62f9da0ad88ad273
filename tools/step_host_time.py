"""Host time per datum of SMCUpdater.update (N = 1e7 precession, no resamples): the whole call, the C entry point inside
it, and what is left for Python -- with qsmc_step (default) and on the round-2 path (QSMC_NO_STEP=1)."""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch  # noqa: E402
import qinfer_amd as qi  # noqa: E402
import qinfer_amd.smc as _smc  # noqa: E402

if os.environ.get("QSMC_NO_STEP"):            # (this tool's own switch: the Python per-datum path, a module attribute since round 6)
    _smc._NO_STEP = True

warnings.simplefilter('ignore')
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
upd = qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0, 1]), device_rng=True, seed=0,
                    resample_thresh=0.0)
eng = upd._eng
ts = [np.array([0.5 + 0.01 * k]) for k in range(64)]
for k in range(50):
    upd.update(k & 1, ts[k & 63])
acc = {"c": 0.0, "calls": 0}
name = "_qsmc_step" if upd._st is not None else None
if name:
    real = eng._qsmc_step

    def timed(*a):
        t0 = time.perf_counter()
        r = real(*a)
        acc["c"] += time.perf_counter() - t0
        acc["calls"] += 1
        return r
    eng._qsmc_step = timed
else:
    real = eng.lib.qsmc_update_fused

    class L:
        def __getattr__(self, k):
            return getattr(eng_lib, k)
    eng_lib = eng.lib

    def timed(*a):
        t0 = time.perf_counter()
        r = real(*a)
        acc["c"] += time.perf_counter() - t0
        acc["calls"] += 1
        return r
    proxy = L()
    proxy.__dict__["qsmc_update_fused"] = timed
    eng.lib = proxy
K = 400
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    upd.update(k & 1, ts[k & 63])
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("path %s  N=%d  per datum: update %.2f us, C call %.2f us, Python around it %.2f us (timer overhead ~0.15 us included)" % (
    "qsmc_step" if name else "round-2 (QSMC_NO_STEP)", n, wall / K * 1e6, acc["c"] / K * 1e6, (wall - acc["c"]) / K * 1e6))
