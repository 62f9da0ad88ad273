#!/usr/bin/env python3
"""config-5 share through batch_update (round 5: sparse-row windows, k_update_multi_tomo) against the per-datum loop and
against the dense window (QSMC_TOMO_DENSE_UPDATE=1 in a second process): p-u/s and the window kernel's time."""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-qinfer_amd"))
import qinfer_amd as qi  # noqa: E402
from qinfer_amd.engine import get_engine  # noqa: E402
import torch  # noqa: E402

eng = get_engine()
basis = qi.tomography.pauli_basis(2)
tm = qi.TomographyModel(basis)
n, K = 1_250_000, 60
rs = np.random.RandomState(0)
np.random.seed(0)
x0 = qi.GinibreDistribution(basis).sample(n)
eps = np.zeros((K,), dtype=tm.expparams_dtype)
for k in range(K):
    eps["meas"][k, 0], eps["meas"][k, rs.randint(1, 16)] = 1, 1
outs = rs.randint(0, 2, K)


class Fixed(qi.Distribution):
    n_rvs = 16

    def sample(self, n=1):
        return x0


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for interval in (5, 8):
        for fast in (True, False):
            upd = qi.SMCUpdater(tm, n, Fixed(), device_rng=True, seed=0)
            upd._batch_fast_path = fast
            upd.batch_update(outs, eps, resample_interval=interval)
            upd.reset()
            torch.cuda.synchronize()
            eng.set_profiling(1)
            upd.batch_update(outs, eps, resample_interval=interval)
            torch.cuda.synchronize()
            ms, tags = eng.profile_read()
            eng.set_profiling(0)
            upd.reset()
            rc0 = upd.resample_count
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            upd.batch_update(outs, eps, resample_interval=interval)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            win = ms[tags == 10]
            one = ms[tags == 0]
            print("interval %d %-14s %.3e p-u/s  %.4f ms/datum  resamples %d  window kernel %s us (%d)  single-datum kernel %s us (%d)  dense=%s" % (
                interval, "windows" if fast else "per-datum loop", n * K / wall, wall / K * 1e3, upd.resample_count - rc0,
                "%.1f" % (win.mean() * 1e3) if len(win) else "-", len(win), "%.1f" % (one.mean() * 1e3) if len(one) else "-", len(one),
                "tomo_dense" in os.environ.get("QSMC_TEST_HOOKS", "")), flush=True)
            del upd
if "tomo_dense" not in os.environ.get("QSMC_TEST_HOOKS", ""):
    import subprocess
    subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, QSMC_TEST_HOOKS="tomo_dense=1"))
