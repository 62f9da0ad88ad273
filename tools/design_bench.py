"""bayes_risk / expected_information_gain of binomial experiments: per-experiment wall time and the design kernel's
time (HIP events), plus the per-outcome sums as an .npy for A/B between kernels (QSMC_HYP_CHAIN1 / QSMC_HYP_NO_CHAIN /
QSMC_HYP_NH).  usage: design_bench.py [N] [out.npy]"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-qinfer_amd"))
import torch  # noqa: E402
import qinfer_amd as qi  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = qi.BinomialModel(qi.SimplePrecessionModel())
upd = qi.SMCUpdater(m, n, qi.UniformDistribution([0, 1]), device_rng=True, seed=0)
ep = np.empty((1,), dtype=m.expparams_dtype)
ep['x'], ep['n_meas'] = 7.0, 25
upd.update(11, ep)
eng = upd._eng
keep = []
for n_meas in (25, 12, 60, 200):
    design = np.empty((4,), dtype=m.expparams_dtype)
    design['x'], design['n_meas'] = [3.0, 9.0, 14.0, 21.0], n_meas
    for name, fn in (("bayes_risk", upd.bayes_risk), ("eig", upd.expected_information_gain)):
        fn(design[:1])
        torch.cuda.synchronize()
        eng.set_profiling(1)
        t0 = time.perf_counter()
        val = fn(design)
        wall = time.perf_counter() - t0
        ms, tags = eng.profile_read()
        eng.set_profiling(0)
        ms, tags = np.asarray(ms), np.asarray(tags)
        per_exp = ms.sum() / 4 * 1e3 if len(ms) else float("nan")
        print("n_meas %3d %-10s wall %.3f ms/experiment, design kernels %.1f us/experiment in %d launches; value[0] %.15g"
              % (n_meas, name, wall / 4 * 1e3, per_exp, len(ms) // 4, val[0]), flush=True)
        keep.append(np.asarray(val))
    for sums in upd._hyp_sums(design[:2]):
        keep.append(np.asarray(sums).ravel())
if len(sys.argv) > 2:
    np.save(sys.argv[2], np.concatenate(keep))
